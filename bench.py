#!/usr/bin/env python3
"""bench.py -- images/sec of the MI355X Faster R-CNN hot path (BASELINE.json metric).

One "step" = one batch of same-size images per GPU through the whole device chain: image (already in HBM) -> backbone head
(f32 MFMA implicit-GEMM / Winograd convs) -> RPN -> proposal layer (decode / clip / top-N select / sort / NMS) -> crop_and_resize
-> per-RoI tail -> cls/bbox -> per-class NMS + top-100 -> detection records in HBM.  `value` counts IMAGES.
N > 1: one process per GPU (torchrun), every rank its own images, fixed-size detection records all-gathered over RCCL/xGMI
every step (north_star); weak scaling.

    python bench.py --gpus 1 --steps 20 --warmup 3                      # the headline: configs[1]
    python bench.py --config c3|c4|c5 ...                                # the other BASELINE.json configs (committed lines in profiles/)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement; extra objects `roofline`, `stages`, `cpu_baseline`).
/root/reference is never read here.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tf-faster-rcnn_amd")
for p in (ROOT, os.path.join(ROOT, "oracle"), PKG, os.path.join(PKG, "lib")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "images/sec (600×1000) ResNet-101 Faster R-CNN at 1/2/4/8 MI355X"
X3_PEAK_TFLOPS = 416.7          # 2500 TFLOP/s dense bf16 MFMA / 6 MFMAs per f32 product block (csrc/gemm_x3.hip)
H2_PEAK_TFLOPS = 833.3          # 2500 TFLOP/s dense fp16 MFMA / 3 MFMAs per f32 product block (csrc/gemm_h2.hip)
F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0                 # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

# BASELINE.json configs; gflop_ref = the reference graph's direct-convolution FLOPs per image
CONFIGS = {
    "c2": dict(label="configs[1]: ResNet-101 VOC 600x1000, 300 proposals, 21 classes, A=9, TEST.MODE nms", net="res101", H=600, W=1000,
               scales=(8, 16, 32), classes=21, post=300, gflop_ref=622.29, batch=8, streams=3),
    "c3": dict(label="configs[2]: ResNet-101 COCO 800x1333, 1000 proposals, 81 classes, A=15, TEST.MODE nms", net="res101", H=800, W=1333,
               scales=(2, 4, 8, 16, 32), classes=81, post=1000, gflop_ref=1787.9, batch=4, streams=3),
    "c4": dict(label="configs[3]: MobileNet-V1 1.0 COCO 600x1000, 300 proposals, 81 classes, A=12", net="mobile", H=600, W=1000,
               scales=(4, 8, 16, 32), classes=81, post=300, gflop_ref=70.2, batch=8, streams=3),
    "c1": dict(label="configs[0] shape on the device chain: VGG16 VOC 600x1000, 300 proposals, 21 classes, A=9", net="vgg16", H=600, W=1000,
               scales=(8, 16, 32), classes=21, post=300, gflop_ref=451.1, batch=4, streams=3),
    "c5": dict(label="configs[4]: ResNet-152 COCO trainval step 600x1000 (anchor_target + proposal_target + losses + backward + "
                     "Momentum SGD), 81 classes, A=12, 256 RoIs", net="res152", H=600, W=1000, scales=(4, 8, 16, 32), classes=81, post=2000,
               gflop_ref=1910.0, batch=1, streams=1),
}
IM_SCALE = 1.6
ANCHOR_RATIOS = (0.5, 1, 2)


def synth_image(c, seed):
    from model.config import cfg
    rng = np.random.RandomState(seed)
    return (rng.rand(1, c["H"], c["W"], 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)


def make_net(c):
    if c["net"] == "res101":
        from nets.resnet_v1 import resnetv1
        return resnetv1(num_layers=101)
    if c["net"] == "res152":
        from nets.resnet_v1 import resnetv1
        return resnetv1(num_layers=152)
    if c["net"] == "vgg16":
        from nets.vgg16 import vgg16
        return vgg16()
    from nets.mobilenet_v1 import mobilenetv1
    return mobilenetv1()


def calibrate_rpn(sess, net, img_d, im_info):
    """(RPN heads and the class head.)  Random-init RPN heads give near-constant scores and ~0 deltas, so NMS collapses the 6000 candidates to ~140 boxes.
    Rescale the two 1x1 RPN heads so the proposal stage sees the statistics SURVEY.md 8d prescribes (logit spread ~1,
    deltas ~N(0,0.2^2)) and the full post_nms_topN proposals survive, as they do with a trained model.  Weights only; the
    architecture and every shape stay those of the reference."""
    with torch.cuda.stream(sess.stream):
        p = net.forward_device(sess, img_d, im_info, use_graph=False)
        sess.stream.synchronize()
        s_cls = float(p["rpn_cls_score"].std().item())
        s_box = float(p["rpn_bbox_pred"].std().item())
        s_head = float(p["cls_score"].std().item())
    scope = net._scope
    sess.variables[scope + "/rpn_cls_score/weights"] *= np.float32(1.0 / max(s_cls, 1e-12))
    sess.variables[scope + "/rpn_bbox_pred/weights"] *= np.float32(0.2 / max(s_box, 1e-12))
    # class logits of spread ~2.5: random-init heads on a random backbone saturate the softmax (scores of exactly 1.0 that tie at the
    # max_per_image cut -- 303 "detections" for the 81-class config in round 2); a trained head's scores are distinct
    sess.variables[scope + "/cls_score/weights"] *= np.float32(2.5 / max(s_head, 1e-12))
    sess.packed.clear()
    sess.graphs.clear()


class Telemetry(object):
    """Shader clock / socket power sampled from the amdgpu sysfs nodes while a region runs (a thread reading a few small files every
    100 ms: no subprocess, nothing on the device).  The node shows every card of the machine, also those other tenants run on: the card
    of THIS process is found by its PCI address (`pci` = "domain:bus:device" of the torch device); only when that fails, the card drawing
    the most power is taken.  `other_cards_max_w` = the busiest OTHER card during the region (context for a reader: the node is shared;
    profiles/r05_t shows the batch-1 latency at 4.00 ms beside neighbours at 1.1-1.4 kW, so their load alone does not explain the 4.95 ms
    some boxes give).  None when the nodes are not there."""

    def __init__(self, pci=None, root="/sys/class/drm"):
        import glob
        self.cards, self.own = [], None
        for sclk in sorted(glob.glob(os.path.join(root, "card[0-9]*", "device", "pp_dpm_sclk"))):
            d = os.path.dirname(sclk)
            hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")))
            fq = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*", "freq1_input")))
            if pci is not None and os.path.basename(os.path.realpath(d)).lower().startswith(pci.lower()):
                self.own = len(self.cards)
            self.cards.append((sclk, hw[0] if hw else None, fq[0] if fq else None))
        self._stop = False

    @staticmethod
    def pci_of(dev):
        """"dddd:bb:dd" of a torch device (None when this torch build does not say)."""
        try:
            p = torch.cuda.get_device_properties(dev)
            return "%04x:%02x:%02x" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
        except Exception:
            return None

    @staticmethod
    def _read(card):
        sclk, power, freq = card
        mhz, watt = None, None
        try:
            if freq is not None:
                mhz = float(open(freq).read().strip()) / 1e6
            else:
                for line in open(sclk):
                    if "*" in line:
                        mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            pass
        try:
            watt = float(open(power).read().strip()) / 1e6
        except Exception:
            pass
        return mhz, watt

    def run(self, fn):
        import threading
        if not self.cards:
            fn()
            return None
        samples, self._stop = [], False

        def loop():
            while not self._stop:
                samples.append([self._read(c) for c in self.cards])
                time.sleep(0.1)
        t = threading.Thread(target=loop, daemon=True)
        t.start()
        try:
            fn()
        finally:
            self._stop = True
            t.join()
        tail = samples[len(samples) // 3:] or samples              # the clock settles after the first third
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        best = None
        for ci in range(len(self.cards)):
            w = med([s[ci][1] for s in tail if s[ci][1]])
            if w is not None and (best is None or w > best[1]):
                best = (ci, w)
        ci = self.own if self.own is not None else (best[0] if best else 0)
        mhz = med([s[ci][0] for s in tail if s[ci][0]])
        watt = med([s[ci][1] for s in tail if s[ci][1]])
        others = [med([s[cj][1] for s in tail if s[cj][1]]) for cj in range(len(self.cards)) if cj != ci]
        others = [w for w in others if w is not None]
        return {"sclk_mhz": None if mhz is None else round(mhz), "socket_w": None if watt is None else round(watt, 1), "samples": len(tail),
                "cards_seen": len(self.cards), "card": "by PCI address" if self.own is not None else "the one drawing the most power",
                "other_cards_max_w": round(max(others), 1) if others else None}


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(variables, image, c):
    """Host-CPU number beside the GPU one (rank 0, N = 1): the oracle's restatement of the SAME workload, one image per run,
    best of 3 after one warm-up run.  Dense part: torch-CPU float32 on the PHYSICAL cores (TensorFlow-CPU is not installable
    offline; over-subscribing the SMT threads halves this leg).  Detection part: the pinned numpy/C oracle, single thread like
    the reference under the GIL; the NMS inside it is the reference's own compiled Cython cpu_nms when oracle/_ref travelled
    with the snapshot (it is built from /root/reference/lib/nms/cpu_nms.pyx by oracle/build_ref.py)."""
    import importlib.util
    import frcnn_oracle as ora
    from dense_ref import DenseRef
    cores = physical_cores()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    ref_nms = None
    try:                                                  # the reference's Cython kernel, loaded by FILE (the package name `nms` is this repo's mirror)
        import glob
        so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "nms", "cpu_nms*.so"))
        if so:
            spec = importlib.util.spec_from_file_location("nms.cpu_nms", so[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            ref_nms = mod.cpu_nms
    except Exception:
        ref_nms = None
    saved_nms = ora.cpu_nms
    if ref_nms is not None:
        ora.cpu_nms = lambda d, t: list(ref_nms(np.ascontiguousarray(d, dtype=np.float32), float(t)))
    ref = DenseRef(variables, 101, c["classes"], c["scales"], ANCHOR_RATIOS, dtype=torch.float32)
    im_info = np.array([c["H"], c["W"], IM_SCALE], dtype=np.float32)
    orig = (int(c["H"] / IM_SCALE), int(c["W"] / IM_SCALE), 3)
    best, parts, best_threads = None, None, cores
    try:
        # the dense leg does not scale to every core of a 2-socket host: try the physical-core count and fractions of it, one
        # warm-up run (weight conversion, thread pool, page faults) then two timed runs each, and report the best
        for threads in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 16)}, reverse=True):
            torch.set_num_threads(threads)
            for it in range(3):
                t0 = time.time()
                with torch.no_grad():
                    out = ref.test_image(image, im_info, post=c["post"])
                t_fwd = time.time() - t0
                t1 = time.time()
                sc, boxes = ora.im_detect_post(out["cls_prob"].astype(np.float32), out["bbox_pred"].astype(np.float32),
                                               out["rois"].astype(np.float32), IM_SCALE, orig)
                ora.test_net_post(sc, boxes, c["classes"])
                t_post = time.time() - t1
                if it > 0 and (best is None or t_fwd + t_post < best):
                    best, parts, best_threads = t_fwd + t_post, (t_fwd, t_post), threads
    finally:
        ora.cpu_nms = saved_nms
        torch.set_num_threads(old_threads)
    cores_phys, cores = cores, best_threads
    return {"value": round(1.0 / best, 4), "unit": "images/sec", "cores": int(cores), "kind": "port",
            "sample": "1 image %dx%d (%d RoIs), best of 2 timed runs after a warm-up run at the best of {1, 1/2, 1/4, 16} x %d physical "
                      "cores: torch-CPU f32 dense restatement on %d threads %.2fs (incl. numpy/C proposal_layer) + per-class NMS %.3fs; "
                      "NMS kernel: %s" % (c["H"], c["W"], c["post"], cores_phys, cores, parts[0], parts[1],
                         "the reference's Cython cpu_nms (oracle/_ref)" if ref_nms is not None else "oracle_c.c restatement")}


def resident_blobs(layer, dev, pool=8):
    """The bench contract times the step with its inputs already in HBM: `pool` synthetic (image, gt) blobs are generated and uploaded
    once, then cycled -- the generator's 1.8 M random numbers per image (~20 ms of host time) are not part of a training step."""
    blobs = []
    for _ in range(pool):
        b = next(layer)
        blobs.append({"data": torch.from_numpy(b["data"]).to(dev), "im_info": b["im_info"], "gt_boxes": torch.from_numpy(b["gt_boxes"]).to(dev)})
    torch.cuda.synchronize()
    i = 0
    while True:
        yield blobs[i % pool]
        i += 1


def train_bench(args, c, dev, world, rank, dist):
    """configs[4]: one SGD step per image per GPU; gradients all-reduced over RCCL when world > 1 (frcnn_hip/parallel.py)."""
    from frcnn_hip.runtime import Session
    from model.config import cfg
    from model.train_val import SolverWrapper, synthetic_data_layer
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False      # experiments/cfgs/res101.yml
    sess = Session(device=dev, seed=cfg.RNG_SEED)
    net = make_net(c)
    net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=ANCHOR_RATIOS)
    sess.init_variables(net.variable_specs())
    ar = None
    if world > 1 or args.dp_constrained:
        from frcnn_hip import parallel
        if world == 1:                                    # a one-rank RCCL group: the same torch.distributed / RCCL calls an N-GPU run makes
            import torch.distributed as dist1
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            dist1.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
        ar = parallel.make_grad_all_reduce()
    sw = SolverWrapper(sess, net, resident_blobs(synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED + rank, image_gain=1 / 256.0), dev),
                       all_reduce=ar, world_size=world, force_dp=args.dp_constrained)
    sw.train_model(max(args.warmup, 1), verbose=False)
    extra = 0
    while cfg.HIP.TRAIN_REPLAY and int(cfg.HIP.TRAIN_PICK_STREAMS) > 0 and getattr(sess, "picked_streams", None) is None and extra < 400:
        sw.train_model(1, verbose=False)                    # (the stream picker times real steps: warm-up lasts until it has chosen)
        extra += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ar_host0 = (getattr(ar, "host_s", 0.0), getattr(ar, "sends", 0))
    t0 = time.perf_counter()
    sw.train_model(args.steps, verbose=False)
    sess.host_enqueue_s = sw.enqueue_s                      # the host's share: launches enqueued, GPU not yet waited for (rounds 3-4 stopped
    # this clock after train_model's final loss read-back, i.e. after a device synchronisation: it read the step time)
    sess.all_reduce_host = None if ar is None else (getattr(ar, "host_s", 0.0) - ar_host0[0], getattr(ar, "sends", 0) - ar_host0[1])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # matrix work of ONE step per matrix pipe (an extra, untimed step with the host-side ledgers on): forward pass from the launch tags
    # (Session.mark), reverse sweep from TrainState.count_flops
    sess.flops_by_pipe, sw.state.flop_ledger = {}, {}
    sess.replay_stats = dict(net.replay_stats)
    replay_on, cfg.HIP.TRAIN_REPLAY = cfg.HIP.TRAIN_REPLAY, False          # (the ledgers are host bookkeeping of the Python step: a replayed step runs none)
    sw.train_model(1, verbose=False)
    cfg.HIP.TRAIN_REPLAY = replay_on
    torch.cuda.synchronize()
    sess.step_flops_by_pipe = {k: sess.flops_by_pipe.get(k, 0) + sw.state.flop_ledger.get(k, 0) for k in ("h2", "x3", "f32")}
    sess.flops_by_pipe, sw.state.flop_ledger = None, None
    sess.dp_note = None
    if args.dp_constrained and world == 1:
        import torch.distributed as dist1
        sess.dp_note = "one replica under the data-parallel rules: bucketed all-reduce (64 MiB) issued from inside the sweep over a one-rank RCCL " \
                       "group (backend %s), ordered after both filter-gradient streams; the solver updates behind each reduced bucket" % dist1.get_backend()
        dist1.destroy_process_group()
    return elapsed, sess


def exchange_records(rec, count_i32, gathered, rank, stamp=None):
    """The data-path collective of one step: the batch's fixed-size detection records [B][dets | count | stamp] of
    this rank are all-gathered (RCCL over xGMI on GPUs, gloo in the CPU test).  stamp (the untimed self-check step only: it costs a
    host-to-device copy): (rank, stamp) goes into two padding floats so that check_exchange can tell WHOSE record of WHICH step sits
    in every slot."""
    from frcnn_hip import parallel
    parallel.set_count(rec, count_i32)
    if stamp is not None:
        parallel.set_stamp(rec, rank, stamp)
    parallel.all_gather_records(rec, gathered)


def check_exchange(rec, gathered, rank, world, step_no):
    """One untimed verification of what the timed steps do: slot r of the gathered tensor must be rank r's record of THIS step
    (stamp), this rank's own slot must equal what it sent bit for bit, and every count must fit the record.  Raises on mismatch."""
    from frcnn_hip import parallel
    g = gathered.detach().cpu()
    mine = rec.detach().cpu()
    if not torch.equal(g[rank], mine):
        raise RuntimeError("rank %d: its own slot of the all-gather differs from the record it sent" % rank)
    flat = g.reshape(world, -1, parallel.REC_FLOATS)
    for r in range(world):
        for b in range(flat.shape[1]):
            st = parallel.get_stamp(flat[r, b])
            if st != (r, step_no):
                raise RuntimeError("rank %d: slot %d image %d carries stamp %r, expected (%d, %d)" % (rank, r, b, st, r, step_no))
            n = float(flat[r, b, parallel.REC_ROWS * 6])
            if not (0 <= n <= 16777216 and n == int(n)):
                raise RuntimeError("rank %d: slot %d image %d has detection count %r" % (rank, r, b, n))
    return True


def run_timed(step, steps, warmup, dist=None, sync=None, between=None):
    """The bench contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps between barrier + device-synchronize
    pairs; returns the seconds of this rank (the caller takes the max over ranks).  `sync` = torch.cuda.synchronize on GPUs (a no-op in
    the CPU test), `between` = per-step host hook (eager mode resets the profile list)."""
    sync = sync or (lambda: None)
    for k in range(warmup):
        step(k)
        if between:
            between()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
        if between:
            between()
    run_timed.enqueue_s = time.perf_counter() - t0          # the host's share: every step enqueued, the device not yet waited for
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    return time.perf_counter() - t0


def child_line(extra, timeout_s):
    """One short run of this script in a child process (its own cfg globals / Session / graphs), returns its parsed JSON line or an
    {"error": ...} record.  Used by the default N = 1 invocation to put the other BASELINE configs and the batch-1 latency into the
    driver-observed line; the parent's timed region is over by then and its buffers simply stay resident beside the child's."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-f32-variant", "--no-other-configs"] + list(extra)
    t0 = time.time()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, env=dict(os.environ, MASTER_PORT=str(29500 + os.getpid() % 2000 + 1)))
    except subprocess.TimeoutExpired:
        return {"error": "timeout after %ds" % timeout_s}
    lines = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr.decode("utf-8", "replace")[-300:])}
    d = json.loads(lines[-1])
    d["wall_s"] = round(time.time() - t0, 1)
    return d


def other_configs(budget_s, t_start):
    """BASELINE.json's other configs (configs[0] as the same VGG16 workload on the device, configs[2], [3], [4]) for 5 steps each, the
    training step once more under the data-parallel rules, and configs[1] with one image on one chain (latency) -- each a child run of
    this script, value / ms per step / per-pipe roofline fraction copied from its line.  Stops adding runs once `budget_s` seconds of
    wall clock have passed since t_start (the skipped ones are named)."""
    runs = [("latency_batch1", ["--config", "c2", "--batch", "1", "--streams", "1", "--steps", "40", "--warmup", "40", "--profile-steps", "0", "--warm-until-stable"]),   # (a 0.1 s run on an idle GPU
            # is timed before the clock has ramped: 5.1 vs 4.3 ms with 5 warm-up steps on two boxes)
            ("c3", ["--config", "c3", "--steps", "5", "--warmup", "3", "--profile-steps", "1"]),
            ("c5", ["--config", "c5", "--steps", "5", "--warmup", "3"]),
            ("c5_dp_constrained", ["--config", "c5", "--steps", "5", "--warmup", "3", "--dp-constrained"]),
            ("c4", ["--config", "c4", "--steps", "5", "--warmup", "3", "--profile-steps", "1"]),
            ("c1", ["--config", "c1", "--steps", "5", "--warmup", "3", "--profile-steps", "1"])]
    out = {}
    for name, extra in runs:
        if time.time() - t_start > budget_s:
            out[name] = {"skipped": "wall-clock budget of %ds for the appended runs used up" % budget_s}
            continue
        d = child_line(extra, 90)
        if "error" in d:
            out[name] = d
            continue
        rf = d.get("roofline") or {}
        rec = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"],
               "images_per_step": d["config"].get("images_per_gpu_per_step", 1), "chains": d["config"].get("chains_in_flight_per_gpu", 1),
               "roofline_bound": rf.get("bound"), "roofline_frac": rf.get("frac"), "mfma_frac": rf.get("mfma_frac", rf.get("frac")), "hbm_frac": rf.get("hbm_frac"),
               "pipes": {k: (v.get("frac_of_pipe_peak") if "frac_of_pipe_peak" in v else v.get("share_of_launched_flops")) for k, v in (rf.get("pipes") or {}).items()},
               "wall_s": d["wall_s"]}
        if d.get("telemetry"):
            rec["sclk_mhz"], rec["socket_w"] = d["telemetry"].get("sclk_mhz"), d["telemetry"].get("socket_w")
            rec["other_cards_max_w"] = d["telemetry"].get("other_cards_max_w")
        if "warm_windows_before_the_timed_one" in d:
            rec["warm_windows"] = d["warm_windows_before_the_timed_one"]
        for k in ("data_parallel_rules", "host_enqueue_ms_per_step", "launch", "all_reduce_host_ms_per_step"):
            if k in d["config"]:
                rec[k] = d["config"][k]
        out[name] = rec
    return out


def train_roofline(by_pipe, step_s, gflop_ref):
    """Roofline object of the training step, priced like the inference line: frac = sum over launches (FLOPs_i / dense peak of the pipe
    launch i issues on) / step time -- the fraction of the matrix pipes' peak the issued instruction mix reaches over the WHOLE timed step
    (forward, losses, reverse sweep, solver; there is no per-launch event pass for the sweep, so the denominator is the step, not the sum
    of the GEMM launches).  `achieved` / `peak` are the same ratio in f32-equivalent TFLOP/s."""
    peaks = {"h2": H2_PEAK_TFLOPS, "x3": X3_PEAK_TFLOPS, "f32": F32_MFMA_PEAK_TFLOPS}
    total = float(sum(by_pipe.values()))
    t_peak = sum(by_pipe[k] / (peaks[k] * 1e12) for k in by_pipe)
    ceiling = total / t_peak / 1e12 if t_peak > 0 else None
    return {"bound": "mfma", "achieved": round(total / step_s / 1e12, 2), "peak": None if ceiling is None else round(ceiling, 1), "unit": "TFLOP/s",
            "frac": round(t_peak / step_s, 4), "traffic": None,
            "gflop_per_step_launched": round(total / 1e9, 1), "gflop_per_step_reference_graph": gflop_ref,
            "pipes": {k: {"share_of_launched_flops": round(v / total, 4), "gflop_per_step": round(v / 1e9, 1),
                          "seconds_at_pipe_peak_per_step": round(v / (peaks[k] * 1e12), 6)} for k, v in by_pipe.items() if v},
            "kernel": "whole training step (forward k_gemm_h2 / k_conv_igemm, reverse sweep: data gradients on the same kernels, filter gradients "
                      "k_wgrad_h2 / k_wgrad_tn): launched f32-equivalent FLOPs per pipe / that pipe's dense peak (833.3 h2, 157.3 f32), over the "
                      "timed step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json config (default c2 = configs[1], the metric's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps of the HIP-event pass that feeds `roofline` / `stages`")
    ap.add_argument("--batch", type=int, default=0, help="images per launch chain: the launches of B images are shared (fills the 256 "
                    "CUs on the 38x63 layers); a step still counts images.  0 = the config's default")
    ap.add_argument("--streams", type=int, default=0, help="launch chains in flight per GPU (independent HIP streams + graphs)")
    ap.add_argument("--reference-order", action="store_true",
                    help="keep the reference op order crop -> 1x1 convs at the block4 entry (default: the 1x1 convs run on the "
                         "feature map and their outputs are cropped; same result up to f32 rounding, 64.5 GFLOP less)")
    ap.add_argument("--mfma", choices=["h2", "x3", "f32"], default="h2",
                    help="h2 (default = cfg.HIP.MFMA_H2 + MFMA_X3): plain GEMMs with Cin, Cout %% 128 == 0 on the fp16 matrix pipe with "
                         "block-scaled two-piece f32 operands (csrc/gemm_h2.hip: three fp16 MFMAs per f32 product, f32 accumulate), the "
                         "other large plain GEMMs on the bf16 pipe with exact 3-way splits (csrc/gemm_x3.hip: six MFMAs), everything else on "
                         "v_mfma_f32_32x32x2_f32; the x3-only and all-f32-MFMA variants are then timed in the same run (`x3_variant`, "
                         "`f32_mfma_variant`).  x3: MFMA_H2 off.  f32: every product on v_mfma_f32_32x32x2_f32")
    ap.add_argument("--hip", action="append", default=[], metavar="KEY=VALUE", help="cfg.HIP[KEY] = VALUE (a Python literal) for this run: the A/B switch "
                    "of every measurement under profiles/ (rounds 2-4 had one flag per experiment; scratch/README.md maps the old names)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="frcnn_set_tuning(KEY, VALUE): thread-local measurement overrides of the library "
                    "(5 second-slot stagger, 6 short-K GEMMs on k_gemm_stream, 7 split-K workgroup target); d:KEY=VALUE = frcnn_detect_set_tuning")
    ap.add_argument("--warm-until-stable", action="store_true", help="repeat the timed window until two consecutive windows agree within 1 %% before the one that counts (the latency child run)")
    ap.add_argument("--pick-streams", type=int, default=-1, help="cfg.HIP.TRAIN_PICK_STREAMS: pool size of the stream picker of the replayed training step (0 = inherit the streams)")
    ap.add_argument("--no-train-replay", action="store_true", help="cfg.HIP.TRAIN_REPLAY False: every training step enqueued by the Python code (c5 A/B)")
    ap.add_argument("--no-f32-variant", action="store_true", help="skip the extra timed regions (x3-only / all-f32-MFMA variants)")
    ap.add_argument("--dp-constrained", action="store_true", help="c5: ONE replica under the data-parallel rules -- at most one filter-gradient side "
                    "stream, the bucketed all-reduce issued from inside the reverse sweep over a one-rank RCCL group, no captured sweep: the step "
                    "every GPU of an N-GPU run executes, timed at N = 1")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configs / the batch-1 latency run that the "
                    "default invocation appends as `other_configs` / `latency_ms_batch1`")
    ap.add_argument("--layer-report", default=None, help="write a per-layer table of the event pass to this file")
    ap.add_argument("--skip-calls", default="", help="MEASUREMENT ONLY (the energy ledger, scratch/energy_ledger.py): comma-separated C-ABI entry names "
                    "(prefix match) whose launches are dropped -- the results are WRONG by construction and the line says so (`ablated`)")
    args = ap.parse_args()
    c = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("N>1 must be launched with torch.distributed.run (one process per GPU)")
    # The default N = 1 invocation also reports the other BASELINE configs and the batch-1 latency (child runs of this script).  They
    # run FIRST, before this process creates its GPU context: two processes with live queues on one GPU are time-sliced by the
    # scheduler, which costs a child 5-25 % (measured: 5.30 vs 4.26 ms latency with the parent's context alive).
    appended = None
    if world == 1 and args.config == "c2" and not args.no_other_configs and not args.no_graph and not (args.batch or args.streams):
        appended = other_configs(75, time.time())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")      # a failed / stuck collective tears the process down instead of hanging
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))

    import frcnn_hip
    frcnn_hip.lib()
    from frcnn_hip.runtime import Session
    from model.config import cfg
    skipped = [n for n in args.skip_calls.split(",") if n]
    if skipped:
        real_call = frcnn_hip.call

        ablate_on = [False]              # switched on after one complete un-ablated pass: every buffer then holds realistic values (the
                                         # matrix pipe's power follows its operands), and the graphs are captured again without the launches

        def ablated_call(name, *a):
            if ablate_on[0] and any(name.startswith(n) for n in skipped):
                return None
            return real_call(name, *a)
        frcnn_hip.call = ablated_call
        from frcnn_hip import ops as _ops
        _ops.call = ablated_call

    cfg.HIP.MFMA_H2 = args.mfma == "h2"
    cfg.HIP.MFMA_X3 = args.mfma in ("h2", "x3")
    if args.no_train_replay:
        cfg.HIP.TRAIN_REPLAY = False
    if args.pick_streams >= 0:
        cfg.HIP.TRAIN_PICK_STREAMS = args.pick_streams
    import ast
    for kv in args.hip:
        k, v = kv.split("=", 1)
        if k not in cfg.HIP:
            raise SystemExit("--hip %s: cfg.HIP has no key %r" % (kv, k))
        cfg.HIP[k] = ast.literal_eval(v)
    for kv in args.tune:
        k, v = kv.split("=", 1)
        if k.startswith("d:"):
            frcnn_hip.lib().frcnn_detect_set_tuning(int(k[2:]), int(v))
        else:
            frcnn_hip.lib().frcnn_set_tuning(int(k), int(v))
    cfg.USE_GPU_NMS = False           # the reference's CPU/Cython suppression rule (cpu_nms.pyx:65): the path BASELINE.json pins
    cfg.TEST.RPN_POST_NMS_TOP_N = c["post"]
    B = args.batch or c["batch"]
    S = max(1, args.streams or c["streams"])
    common = {"metric": METRIC, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
              "dtype": {"f32": "f32",
                        "h2": "f32 (operands, accumulators, results; GEMMs with Cin, Cout % 128 == 0 form each f32 product from two-piece "
                              "fp16 operands with an exact power-of-two scale per 128-k block on the fp16 matrix pipe -- 3 MFMAs / product, "
                              "dropped terms <= 3 * 2^-22, measured error vs float64 below the f32-MFMA kernel's; other large GEMMs: exact "
                              "3-way bf16 splits, 6 MFMAs / product; the rest on the f32 MFMA)",
                        "x3": "f32 (operands, accumulators, results; the large GEMMs form each f32 product from exact 3-way bf16 "
                              "operand splits on the bf16 matrix pipe -- 6 MFMAs / product, dropped terms <= 2^-24 -- the rest on the f32 MFMA)"}[args.mfma]}

    if args.config == "c5":
        elapsed, sess = train_bench(args, c, dev, world, rank, dist)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:
            value = world * args.steps / elapsed
            out = dict(common, value=round(value, 3), ms_per_step=round(1000.0 * elapsed / args.steps, 4),
                       config={"workload": c["label"] + "; one image per GPU per step", "parallelism": "dp%d (RCCL gradient all-reduce)" % world,
                               "launch": "%s (forward, reverse sweep, solver), filter gradients on %d side stream(s)" % (
                                   "one recorded launch list per step, replayed (cfg.HIP.TRAIN_REPLAY: %r)" % (sess.replay_stats,)
                                   if cfg.HIP.TRAIN_REPLAY else "eager: every step enqueued by the Python code",
                                   int(cfg.HIP.WGRAD_STREAM)),
                               "host_enqueue_ms_per_step": round(1000.0 * sess.host_enqueue_s / args.steps, 3),
                               "gflop_per_step_reference_graph": c["gflop_ref"]},
                       roofline=train_roofline(sess.step_flops_by_pipe, elapsed / args.steps, c["gflop_ref"]))
            if getattr(sess, "pick_log", None):
                out["config"]["stream_pick"] = {"pool": int(cfg.HIP.TRAIN_PICK_STREAMS), "ms_per_step_by_candidate": [[w, round(1000.0 * t, 3)] for w, t in sess.pick_log]}
            if sess.dp_note:
                out["config"]["data_parallel_rules"] = sess.dp_note
            if sess.all_reduce_host is not None:
                out["config"]["all_reduce_host_ms_per_step"] = round(1000.0 * sess.all_reduce_host[0] / args.steps, 3)
                out["config"]["all_reduce_calls_per_step"] = round(sess.all_reduce_host[1] / float(args.steps), 2)
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    sess = Session(device=dev, seed=cfg.RNG_SEED)
    nets = []
    for i in range(S):                                   # one Network (= one set of static buffers + one hipGraph) per stream
        n_ = make_net(c)
        n_.create_architecture("TEST", c["classes"], tag="s%d" % i, anchor_scales=c["scales"], anchor_ratios=ANCHOR_RATIOS)
        n_._fuse_tail_entry = not args.reference_order
        nets.append(n_)
    net = nets[0]
    sess.init_variables(net.variable_specs())            # weights are shared by all streams
    im_info = np.array([c["H"], c["W"], IM_SCALE], dtype=np.float32)
    orig_shape = (int(c["H"] / IM_SCALE), int(c["W"] / IM_SCALE))
    gain = np.float32(1 / 64.0 if c["net"] == "vgg16" else 1.0)     # VGG has no normalisation: keep 13 random conv layers finite
    image = synth_image(c, cfg.RNG_SEED + rank) * gain   # every rank its own image

    from frcnn_hip import parallel
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    recs, views, counts, gathered, imgs = [], [], [], [], []
    for i in range(S):
        r_, v_ = parallel.new_record(dev, batch=B)       # fixed-size detection records [B][dets | count]
        recs.append(r_); views.append(v_)
        counts.append(torch.zeros((B,), dtype=torch.int32, device=dev))
        gathered.append(torch.zeros((world,) + tuple(r_.shape), dtype=torch.float32, device=dev) if world > 1 else None)
        with torch.cuda.stream(streams[i]):                 # every (rank, stream, slot) its own image, resident in HBM
            batch_np = np.concatenate([synth_image(c, cfg.RNG_SEED + rank + 1000 * i + 100000 * b) * gain for b in range(B)], axis=0)
            imgs.append(nets[i]._stage_image(sess, batch_np))
    torch.cuda.synchronize()
    calibrate_rpn(sess, nets[0], imgs[0][:1].contiguous(), im_info)   # synthetic-data preparation, outside any timed region
    run_stream = streams[0]
    img_d, dets_view, count_i32 = imgs[0], views[0], counts[0]

    def step(k, stamp=None):
        i = k % S
        with torch.cuda.stream(streams[i]):
            nets[i].detect_device(sess, imgs[i], im_info, orig_shape, out=views[i], count=counts[i])
            if world > 1:
                exchange_records(recs[i], counts[i], gathered[i], rank, stamp)

    def timed_region():
        """W untimed warm-up steps (first use of a configuration also builds / captures its graphs), then exactly K steps between
        barrier + synchronize pairs (run_timed)."""
        if args.no_graph:
            sess.profile = []                            # forward_device runs eagerly while profile is not None

        def reset_profile():
            if args.no_graph:
                sess.profile = []
        return run_timed(step, args.steps, max(args.warmup, S), dist, torch.cuda.synchronize, reset_profile)

    if skipped:
        timed_region()
        ablate_on[0] = True
        sess.graphs.clear()
    warm_windows = None
    if args.warm_until_stable:
        # a sub-second run on an idle GPU is timed before the clock has ramped (driver 4.84 ms vs 4.07 ms builder-run in round 4): repeat the
        # whole window until two consecutive ones agree within 1 % (at most 12), then time the one that counts
        prev, warm_windows = None, 0
        trace_n = int(os.environ.get("FRCNN_BENCH_WINDOWS", "0"))      # measurement aid: run this many windows whatever they say, print each
        while warm_windows < max(12, trace_n):
            t = timed_region()
            warm_windows += 1
            if trace_n:
                print("window %d: %.4f ms per step" % (warm_windows, 1000.0 * t / args.steps), file=sys.stderr, flush=True)
            if prev is not None and abs(t - prev) <= 0.01 * prev and warm_windows >= trace_n:
                break
            prev = t
    elapsed = timed_region()
    host_enqueue_s = getattr(run_timed, "enqueue_s", 0.0)       # (of the timed region that counts: a graph launch per step and chain)
    exchange_ok = None
    if world > 1:
        # self-check of the collective, untimed: one more step on every chain, then every rank verifies every slot (a wrong or stale
        # record, or an RCCL failure, ends the run with a non-zero exit instead of a silently wrong number)
        for i in range(S):
            step(i, stamp=1000 + i)
        torch.cuda.synchronize()
        for i in range(S):
            check_exchange(recs[i], gathered[i], rank, world, 1000 + i)
        exchange_ok = True
    n_det = int(count_i32[0].item())
    n_rois = int(net._num_rois[0].item())
    flops_per_image = sess.flops_last_forward / B        # launched MFMA FLOPs (Winograd / commuted crop already taken out)

    # ---- HIP-event pass: events around every launch group, on the stream the kernels run on, over `profile_steps` further
    #      steps of the same workload (a hipGraph replay cannot carry events, so this pass launches eagerly on ONE chain)
    per_layer, conv = {}, [0.0, 0, 0, 0]
    pipes = {"h2": [0.0, 0, 0, 0], "x3": [0.0, 0, 0, 0], "f32": [0.0, 0, 0, 0]}   # ms, f32-equivalent FLOPs, launches, algorithmic bytes per matrix pipe
    if rank == 0 and args.profile_steps > 0:
        with torch.cuda.stream(run_stream):
            sess.profile = []
            for _ in range(args.profile_steps):           # no collective here: only rank 0 runs this pass
                net.detect_device(sess, img_d, im_info, orig_shape, out=dets_view, count=count_i32)
            run_stream.synchronize()
            for tag, fl, e0, e1, nb in sess.profile:
                ms = e0.elapsed_time(e1)
                a = per_layer.setdefault(tag, [0.0, 0, 0, 0])
                a[0] += ms; a[1] += fl; a[2] += 1; a[3] += nb
                if tag.startswith("conv:"):
                    conv[0] += ms; conv[1] += fl; conv[2] += 1; conv[3] += nb
                    pp = pipes["h2" if tag.startswith("conv:h2:") else "x3" if tag.startswith("conv:x3:") else "f32"]
                    pp[0] += ms; pp[1] += fl; pp[2] += 1; pp[3] += nb
            sess.profile = None
        if args.layer_report:
            with open(args.layer_report, "w") as f:
                f.write("# tag  launches  us/launch  GFLOP/launch  TFLOP/s  GB/s(algorithmic)\n")
                for tag, (ms, fl, n, nb) in per_layer.items():
                    f.write("%-70s %3d %9.1f %9.3f %8.1f %8.0f\n" % (tag, n, 1000 * ms / n, fl / n / 1e9, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0,
                                                                     nb / (ms * 1e-3) / 1e9 if ms > 0 else 0))

    f32_variant, x3_variant, telemetry = None, None, None
    if world == 1 and not args.no_graph:
        telemetry = Telemetry(Telemetry.pci_of(dev)).run(lambda: [timed_region() for _ in range(3)])      # clock / power of the shipped configuration
    if args.mfma in ("h2", "x3") and world == 1 and not args.no_f32_variant and not args.no_graph:
        # the same workload on the other pipe choices, timed in the same run (rank 0, N = 1 like cpu_baseline)
        keep = (cfg.HIP.MFMA_H2, cfg.HIP.MFMA_X3)
        if args.mfma == "h2":
            cfg.HIP.MFMA_H2 = False
            e3 = timed_region()
            x3_variant = {"value": round(args.steps * B / e3, 3), "unit": "images/sec", "ms_per_step": round(1000.0 * e3 / args.steps, 4),
                          "what": "identical run with cfg.HIP.MFMA_H2 = False (bench.py --mfma x3): round 2's configuration"}
        cfg.HIP.MFMA_H2 = cfg.HIP.MFMA_X3 = False
        e32 = timed_region()
        cfg.HIP.MFMA_H2, cfg.HIP.MFMA_X3 = keep
        f32_variant = {"value": round(args.steps * B / e32, 3), "unit": "images/sec", "ms_per_step": round(1000.0 * e32 / args.steps, 4),
                       "what": "identical run with cfg.HIP.MFMA_H2 = MFMA_X3 = False (bench.py --mfma f32): all GEMMs on the f32 MFMA"}

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * args.steps * B / elapsed
        out = dict(common, value=round(value, 3), ms_per_step=round(1000.0 * elapsed / args.steps, 4))
        out["config"] = {"workload": c["label"] + "; image in HBM -> <=100 detections in HBM", "images_per_gpu_per_step": B,
                         "chains_in_flight_per_gpu": S, "parallelism": "dp%d (one batch per GPU, all-gather of detection records)" % world,
                         "launch": "eager" if args.no_graph else "hipGraph replay", "rois": n_rois, "detections": n_det,
                         "host_enqueue_ms_per_step": round(1000.0 * host_enqueue_s / args.steps, 3),
                         "nms_rule": "cpu_nms (ovr >= thresh)",
                         "winograd": {"m": int(cfg.HIP.WINOGRAD_M), "f2_scopes": list(cfg.HIP.WINOGRAD_F2_SCOPES),
                                      "direct_scopes": list(cfg.HIP.WINOGRAD_DIRECT_SCOPES), "crops_7x7": bool(cfg.HIP.WINOGRAD_7X7)}
                         if cfg.HIP.WINOGRAD else None,
                         "graph": "reference op order" if args.reference_order else
                                  "block4/unit_1 1x1 convs commuted past the bilinear crop (exact algebra, same outputs to f32 rounding; "
                                  "--reference-order keeps crop -> conv)",
                         "mfma": {"h2": "cfg.HIP.MFMA_H2: plain GEMMs with Cin, Cout %% 128 == 0 and >= %d tiles on v_mfma_f32_32x32x16_f16 with "
                                        "block-scaled two-piece f32 operands (csrc/gemm_h2.hip), operand planes emitted by the producers; "
                                        "remaining large plain GEMMs: cfg.HIP.MFMA_X3 (bf16 pipe, exact 3-way splits); the stem, strided / "
                                        "small-Cout convolutions and heads on v_mfma_f32_32x32x2_f32" % int(cfg.HIP.H2_MIN_TILES),
                                  "x3": "cfg.HIP.MFMA_X3: plain GEMMs with Cout % 128 == 0 and >= 150 tiles on v_mfma_f32_32x32x16_bf16 with "
                                        "exactly split f32 operands (csrc/gemm_x3.hip); the stem, strided / small-Cout convolutions and heads on "
                                        "v_mfma_f32_32x32x2_f32", "f32": "v_mfma_f32_32x32x2_f32 everywhere"}[args.mfma],
                         "gflop_per_image_launched": round(flops_per_image / 1e9, 2), "gflop_per_image_reference_graph": c["gflop_ref"]}
        out["telemetry"] = telemetry
        if skipped:
            out["ablated"] = {"skipped_calls": skipped, "note": "MEASUREMENT ONLY: these launches were dropped, the detections are wrong by construction"}
        if warm_windows is not None:
            out["warm_windows_before_the_timed_one"] = warm_windows
        if exchange_ok is not None:
            out["all_gather_self_check"] = "ok: every rank found every rank's record of the checked step in its slot (untimed extra step per chain); " \
                                           "expected scaling: %d B per image in the all-gather, no other coupling -> linear in N" % (4 * parallel.REC_FLOATS)
        if x3_variant is not None:
            out["x3_variant"] = x3_variant
        if f32_variant is not None:
            out["f32_mfma_variant"] = f32_variant
        if conv[2]:
            steps_p = args.profile_steps
            ach = conv[1] / (conv[0] * 1e-3) / 1e12
            # HBM bytes per GEMM launch and matrix-pipe busy fraction from the committed rocprofv3 PMC passes of THIS command
            # (scratch/pmc_traffic.py; regenerated per round, the file names the commit it was taken at)
            traffic, tsrc, busy = None, None, None
            for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):          # the newest committed pass
                tpath = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tpath) and args.config == "c2" and not args.reference_order and args.mfma == "h2":
                    try:
                        pj = json.load(open(tpath))
                        if int(pj.get("images_per_step", 4)) != B:       # a pass taken at another batch size says nothing about this launch
                            continue
                        traffic, tsrc, busy = round(pj["hbm_bytes_per_launch"]), "profiles/" + name, round(pj["mfma_util_conv_launches"], 4)
                        break
                    except Exception:
                        traffic = None
            # what rocprofv3 counters say binds the conv3-class launch as it is shipped (block4 conv3 alone: residual as planes, planes out,
            # cfg 31; separate --pmc passes, scratch/gpu_r05_v.sh -> the committed JSON): the waves wait on s_waitcnt half of their cycles,
            # the matrix pipe is busy 0.28, write requests stall 0.06 -- latency inside the tile's serial chain, neither pipe nor HBM
            counters = None
            cpath = next((q for q in (os.path.join(ROOT, "profiles", n_) for n_ in ("r06_counters_conv3.json", "r05_counters_conv3.json")) if os.path.exists(q)),
                         os.path.join(ROOT, "profiles", "r05_counters_conv3.json"))
            if os.path.exists(cpath):
                try:
                    cj = json.load(open(cpath))
                    counters = {k: cj[k] for k in ("source", "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES", "SQ_WAIT_ANY/SQ_WAVE_CYCLES", "mfma_busy",
                                                   "TCC_EA0_WRREQ_STALL/TCC_EA0_WRREQ", "SQ_LDS_BANK_CONFLICT", "hbm_bytes_per_launch") if k in cj}
                    counters["reading"] = cj.get("reading", "latency-bound: waves parked on s_waitcnt, neither the matrix pipe nor HBM saturated")
                    if cj.get("hbm_bytes_per_launch") and cj.get("algorithmic_bytes_per_launch"):
                        counters["traffic_ratio"] = round(cj["hbm_bytes_per_launch"] / cj["algorithmic_bytes_per_launch"], 3)
                        counters["algorithmic_bytes_per_launch"] = cj["algorithmic_bytes_per_launch"]
                except Exception:
                    counters = None
            whole_launched = flops_per_image * value / world / 1e12
            peaks = {"h2": H2_PEAK_TFLOPS, "x3": X3_PEAK_TFLOPS, "f32": F32_MFMA_PEAK_TFLOPS}
            t_at_peak = sum(pp[1] / (peaks[k] * 1e12) for k, pp in pipes.items())       # seconds the issued MFMA mix needs at its pipes' peaks
            ceiling = conv[1] / t_at_peak / 1e12                                         # f32-equivalent TFLOP/s of that mix at peak
            # which resource binds, per pipe: the time the pipe's launches need at the matrix pipe's dense peak against the time their
            # algorithmic bytes need at the HBM peak (8 TB/s); the larger one names the bound
            t_hbm = {k: pp[3] / (HBM_PEAK_GBS * 1e9) for k, pp in pipes.items()}
            t_mfma = {k: pp[1] / (peaks[k] * 1e12) for k, pp in pipes.items()}
            bound_by_pipe = {k: ("hbm" if t_hbm[k] > t_mfma[k] else "mfma") for k, pp in pipes.items() if pp[2]}
            all_bytes_step = sum(v[3] for v in per_layer.values()) / steps_p            # every launch group of a step: GEMMs, transforms, crop, ...
            step_s = elapsed / args.steps
            hbm_bound = sum(t_hbm.values()) > t_at_peak
            hbm_ach = conv[3] / (conv[0] * 1e-3) / 1e9                                     # algorithmic GB/s over the GEMM launches
            mfma_frac = t_at_peak / (conv[0] * 1e-3)
            out["roofline"] = {
                # The contract's five fields are the MATRIX-PIPE view of the dominant kernel family (the GEMM launches), as in rounds 1-3:
                # frac = mfma_frac = sum over launches (FLOPs_i / dense peak of the pipe launch i issues on) / sum of their HIP-event times.
                # (Round 4 switched these fields to the HBM view because the launches' algorithmic bytes need more time at 8 TB/s than
                # their MFMAs need at peak; that is a comparison of two FLOORS, each a quarter of the measured time, not a measurement --
                # it is kept below as `floor_comparison`.  What the counters say about the launches is in `counters`.)
                "bound": "mfma",
                "achieved": round(ach, 2), "peak": round(ceiling, 1), "unit": "TFLOP/s", "frac": round(mfma_frac, 4),
                "mfma_frac": round(mfma_frac, 4), "mfma_peak_f32_equivalent": round(ceiling, 1), "hbm_frac": round(hbm_ach / HBM_PEAK_GBS, 4),
                "floor_comparison": {"larger_floor": "hbm" if hbm_bound else "mfma", "by_pipe": bound_by_pipe,
                                     "what": "which ideal time is larger over the GEMM launches: algorithmic bytes at 8 TB/s or issued MFMAs at the "
                                             "pipes' dense peaks (seconds_at_peak_per_step); both are far below the measured time"},
                "counters": counters,
                "seconds_at_peak_per_step": {"mfma": round(t_at_peak / steps_p, 6), "hbm_8TBs_gemm_launches": round(sum(t_hbm.values()) / steps_p, 6),
                                             "hbm_8TBs_all_launches": round(all_bytes_step / (HBM_PEAK_GBS * 1e9), 6), "timed_step": round(step_s, 6)},
                # the HBM side of the GEMM launches and of the whole step:
                "hbm": {"gemm_achieved_GBs": round(conv[3] / (conv[0] * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                        "gemm_frac": round(conv[3] / (conv[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "algorithmic_GB_per_step_all_launches": round(all_bytes_step / 1e9, 3),
                        "hbm_frac_timed_region": round(all_bytes_step / step_s / 1e9 / HBM_PEAK_GBS, 4),
                        "gemm_arithmetic_intensity_flop_per_byte": round(conv[1] / max(conv[3], 1), 1)},
                # mfma_frac = sum_launch (FLOPs_i / peak of the pipe launch i issues on) / sum_launch time_i  (event pass, one chain):
                # the fraction of the matrix pipes' dense peak the issued instruction mix achieves; achieved_f32_equivalent /
                # mfma_peak_f32_equivalent are the same ratio in f32-equivalent TFLOP/s (peak = what this mix of 3-MFMA / 6-MFMA /
                # f32-MFMA products could reach)
                "achieved_f32_equivalent": round(ach, 2),
                "pipe_peaks_f32_equivalent": {"h2 (3 x v_mfma_f32_32x32x16_f16 per product, 2500 / 3)": H2_PEAK_TFLOPS,
                                              "x3 (6 x v_mfma_f32_32x32x16_bf16, 2500 / 6)": X3_PEAK_TFLOPS,
                                              "f32 (v_mfma_f32_32x32x2_f32)": F32_MFMA_PEAK_TFLOPS},
                # the same kernels credited with the reference graph's direct-convolution FLOPs over the f32-MFMA peak of the dtype
                # (SURVEY 8d / BASELINE.md formula): > 1 is what Winograd, the commuted crop and the 16-bit pipes buy
                "frac_algorithmic": round(c["gflop_ref"] * value / world / 1e3 / F32_MFMA_PEAK_TFLOPS, 4),
                # launched FLOPs over the TIMED region's clock (all chains, non-conv stages included), same per-pipe peaks
                "frac_timed_region": round(whole_launched / ceiling, 4),
                "mfma_busy": busy, "traffic": traffic, "traffic_source": tsrc,
                # pipeline average (all GEMM launches) and the DOMINANT launch alone (block4 conv3 as shipped; `counters`): counter bytes /
                # algorithmic bytes -- well above 1 = re-fetched operands, the first thing to fix
                "traffic_ratio_all_gemm_launches": None if not traffic else round(traffic / max(conv[3] // max(conv[2], 1), 1), 3),
                "traffic_ratio_dominant_launch": None if not counters else counters.get("traffic_ratio"),
                "sclk_mhz": None if telemetry is None else telemetry["sclk_mhz"], "socket_w": None if telemetry is None else telemetry["socket_w"],
                # the dense peaks above are quoted at the 2.4 GHz boost clock; every configuration of this pipeline runs at the socket
                # power limit (1 400 W) with the firmware picking the clock (profiles/r04_e_h2_power.txt), so the same fraction against
                # the peak AT THE CLOCK THE RUN HELD is reported too
                "mfma_frac_at_running_clock": None if (telemetry is None or not telemetry["sclk_mhz"]) else round(mfma_frac * 2400.0 / telemetry["sclk_mhz"], 4),
                "kernel": "k_gemm_h2 (fp16 pipe, block-scaled two-piece operands) + k_gemm_x3 (bf16 pipe, exact 3-way split) + k_conv_igemm / "
                          "k_gemm_stream (f32 MFMA 32x32x2), all tile shapes; Winograd GEMMs included",
                "algorithmic_bytes_per_launch": conv[3] // max(conv[2], 1), "launches_per_step": conv[2] // steps_p,
                "pipes": {k: {"share_of_launched_flops": round(pp[1] / conv[1], 4), "launches_per_step": pp[2] // steps_p,
                              "achieved_f32_equivalent": round(pp[1] / (pp[0] * 1e-3) / 1e12, 2),
                              "frac_of_pipe_peak": round(pp[1] / (pp[0] * 1e-3) / 1e12 / peaks[k], 4)} for k, pp in pipes.items() if pp[2]},
                "avg_launch_us": round(1000.0 * conv[0] / conv[2], 2), "conv_ms_per_image": round(conv[0] / steps_p / B, 3)}
            # the bandwidth-bound stages (north_star: achieved HBM GB/s): algorithmic bytes / event time of the stage
            stages = {}
            for tag, key in (("op:proposal_layer", "proposal_layer"), ("op:proposal_layer_tf", "proposal_layer"),
                             ("op:crop_and_resize", "crop_and_resize"), ("op:detect_post", "detect_post"), ("op:wino_in", "winograd_input_transform"),
                             ("op:wino_out", "winograd_output_transform"), ("op:h2_split", "h2_split"), ("op:maxpool", "maxpool"), ("op:spatial_mean", "spatial_mean"),
                             ("op:dwconv3x3", "depthwise_conv")):
                if tag in per_layer:
                    ms, _, n, nb = per_layer[tag]
                    stages[key] = {"us_per_image": round(1000.0 * ms / steps_p / B, 2), "launches_per_step": n // steps_p,
                                   "algorithmic_MB_per_step": round(nb / steps_p / 1e6, 2),
                                   "achieved_GBs": round(nb / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                                   "frac_of_hbm_peak": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
            out["stages"] = stages
        if world == 1 and not args.no_cpu_baseline and c["net"] == "res101":
            out["cpu_baseline"] = cpu_baseline(sess.variables, image, c)
        if appended is not None:
            oc = appended
            lat = oc.pop("latency_batch1", None)
            if lat and "ms_per_step" in lat:
                out["latency_ms_batch1"] = lat["ms_per_step"]
                out["latency_batch1"] = {"images_per_sec": lat["value"], "what": "configs[1], ONE image on ONE chain (hipGraph replay): the time "
                                         "from image in HBM to detections in HBM; warmed until two consecutive windows agree within 1 %",
                                         "steps": lat["steps"], "sclk_mhz": lat.get("sclk_mhz"), "socket_w": lat.get("socket_w"),
                                         "warm_windows": lat.get("warm_windows"), "other_cards_max_w": lat.get("other_cards_max_w"),
                                         "note": "a chain of ~300 dependent graph nodes of ~10 us; two regimes box by box (4.0 / 4.95 ms, DESIGN section 0); "
                                                 "other_cards_max_w = the busiest neighbouring card of the node during the run"}
            elif lat:
                out["latency_batch1"] = lat
            out["other_configs"] = oc
        # LAST on the line on purpose: a log tail that cuts the line from the front still carries the numbers a reader compares first
        # (round 5's driver tail lost `x3_variant`)
        out["summary"] = {"images_per_sec": out["value"], "ms_per_step": out["ms_per_step"],
                          "x3_variant_images_per_sec": None if x3_variant is None else x3_variant["value"],
                          "f32_mfma_variant_images_per_sec": None if f32_variant is None else f32_variant["value"],
                          "roofline_frac": (out.get("roofline") or {}).get("frac"),
                          "traffic_ratio_dominant_launch": (out.get("roofline") or {}).get("traffic_ratio_dominant_launch"),
                          "latency_ms_batch1": out.get("latency_ms_batch1"),
                          "socket_w": None if telemetry is None else telemetry.get("socket_w"),
                          "j_per_image": None if (telemetry is None or not telemetry.get("socket_w")) else round(telemetry["socket_w"] / value, 3)}
        out["j_per_image"] = out["summary"]["j_per_image"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:                     # incl. RCCL errors surfacing as exceptions: non-zero exit, never a hang in teardown
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
