"""Is the ResNet-152 training step bound by the host?  (a) the C5 step on a TINY image (160 x 224, same layers, same launch count, a few
ms of GPU work): what remains is the host's enqueue time per step; (b) cProfile of the host side of full-size steps (tottime top 35);
(c) per-step count of C-ABI calls and of torch stream / event operations.   python scratch/train_host_profile.py [out.txt]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets sys.path for the package)
from frcnn_hip import ops  # noqa: E402
from frcnn_hip.runtime import Session  # noqa: E402
from model.config import cfg  # noqa: E402
from model.train_val import SolverWrapper, synthetic_data_layer  # noqa: E402

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
c = bench.CONFIGS["c5"]


def run(h, w, steps, tag, prof=False):
    sess = Session(device=dev, seed=cfg.RNG_SEED)
    net = bench.make_net(c)
    net.create_architecture("TRAIN", c["classes"], tag=tag, anchor_scales=c["scales"], anchor_ratios=bench.ANCHOR_RATIOS)
    sess.init_variables(net.variable_specs())
    layer = bench.resident_blobs(synthetic_data_layer(c["classes"], seed=3, height=h, width=w, scale=1.0, image_gain=1 / 256.0), dev)
    sw = SolverWrapper(sess, net, layer)
    sw.train_model(4, verbose=False)
    torch.cuda.synchronize()
    calls = {}
    real = ops.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return real(name, *a)
    ops.call = counting
    sw.train_model(1, verbose=False)
    ops.call = real
    torch.cuda.synchronize()
    pr = cProfile.Profile() if prof else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    sw.train_model(steps, verbose=False)
    if pr:
        pr.disable()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    out.write("%s: %dx%d  host enqueue %.2f ms / step, completed %.2f ms / step, %d C-ABI calls / step\n" % (tag, h, w, 1e3 * host / steps, 1e3 * total / steps, sum(calls.values())))
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
        out.write(s.getvalue())
        top = sorted(calls.items(), key=lambda kv: -kv[1])[:25]
        out.write("C-ABI calls per step: " + ", ".join("%s %d" % kv for kv in top) + "\n")


run(160, 224, 20, "tiny")
run(600, 1000, 20, "full")
run(600, 1000, 10, "full_cprofile", prof=True)
out.flush()
