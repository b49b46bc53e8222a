"""frcnn_gemm_h2 on the conv3-CLASS launches of a bench step (8 images per launch): short K, residual + float32 + planes epilogue, and the
fused-mean form of the tail's last unit -- time per tile configuration (interleaved A/B in one process), f32-equivalent TFLOP/s,
algorithmic GB/s, bit equality against cfg 9.  Round 5.

    python scratch/h2_conv3.py 9,21,12[,more] [shape,shape,...] [--lib path/to/ablation.so] [--single N]

--single N: only N launches of the first cfg on the first shape (for rocprofv3 --pmc passes: one kernel, nothing else).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
args = [a for a in sys.argv[1:]]
lib_path, single = None, 0
if "--lib" in args:
    i = args.index("--lib"); lib_path = args[i + 1]; del args[i:i + 2]
if "--single" in args:
    i = args.index("--single"); single = int(args[i + 1]); del args[i:i + 2]
import frcnn_hip
if lib_path:
    frcnn_hip.LIB_PATH = os.path.abspath(lib_path)
import numpy as np, torch
from frcnn_hip import ops
dev = torch.device("cuda:0")
shapes = {  # name: (G, M, N, K, residual, f32 out, planes out, mean rows)
    "b4c3x8":  (1, 117600, 2048, 512, True, True, True, 0),        # block4 units 1-2 conv3: residual + f32 trunk + next conv1's planes
    "b4c3x8m": (8, 14700, 2048, 512, True, False, False, 49),      # block4 unit 3 conv3 + reduce_mean (one batch entry per image)
    "b3c3x8":  (1, 19152, 1024, 256, True, True, True, 0),         # block3 conv3 (23 per step)
    "b2c3x8":  (1, 75000, 512, 128, True, True, True, 0),          # block2 conv3
    "b4c3x8p": (1, 117600, 2048, 512, "planes", False, True, 0),   # ... as shipped since round 5 (cfg.HIP.H2_TRUNK_PLANES): residual read as planes, planes only out
    "b3c3x8p": (1, 19152, 1024, 256, "planes", False, True, 0),
    "b3scx8":  (1, 19152, 1024, 512, False, True, True, 0),        # block3 unit 1 shortcut
    "w7x8":    (121, 2400, 512, 512, False, True, False, 0),       # 7 x 7 Winograd products of the tail's conv2
    "w3x8":    (36, 1280, 256, 256, False, True, False, 0),        # block3 conv2's Winograd products
    "wrpnx8":  (36, 1280, 512, 1024, False, True, False, 0),
    "b4c1x8":  (1, 117600, 512, 2048, False, False, True, 0),      # block4 conv1: planes only (feeds the Winograd input transform? no: f32) -- long K control
    "b3c1x8":  (1, 19152, 256, 1024, False, True, False, 0),
    "cal_k128": (1, 117600, 128, 128, "planes", False, True, 0),    # counter calibration: ONE column tile (the filter planes are 64 KB: no re-fetch), X = residual = out = 60.2 MB
    "b4c3x1":  (1, 14700, 2048, 512, True, True, True, 0),         # single image (latency mode)
    "b3c3x1":  (1, 2394, 1024, 256, True, True, True, 0),
}
cfgs = [int(c) for c in (args[0] if args else "9,21,12").split(",")]
only = args[1].split(",") if len(args) > 1 else ["b4c3x8", "b4c3x8m", "b3c3x8", "b2c3x8", "w7x8", "w3x8", "b4c1x8", "b3c1x8"]
if not single:
    print("%-8s %-4s %9s %9s %8s %8s  %s" % ("shape", "cfg", "med_us", "min_us", "TFLOP/s", "GB/s", "bits vs cfg %d" % cfgs[0]))
for name in only:
    G, M, N, K, has_res, f32o, planes, mean_rows = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 4 - 2)
    shared_w = mean_rows > 0
    w = torch.randn(1 if shared_w else G, N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if (G == 1 or shared_w) else None
    res = torch.randn(G * M, N, device=dev) if has_res else None
    if has_res == "planes":
        res = ops.h2_split(res.clamp(min=0))
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    del x
    outs, outp = {}, {}
    share = G * M * N * 4 > (1 << 29)                    # big outputs: the timed configurations write ONE buffer, bit checks run one by one
    for c in cfgs:
        if mean_rows:
            outs[c] = torch.empty(G * M // mean_rows, N, device=dev)
        else:
            outs[c] = (outs[cfgs[0]] if (share and c != cfgs[0]) else torch.empty(G * M, N, device=dev)) if f32o else None
            outp[c] = (outp[cfgs[0]] if (share and c != cfgs[0]) else ops.H2.empty(G * M, N, dev)) if planes else None

    def run(c):
        if mean_rows:
            return ops.gemm_h2_mean(xp, wp, G, M, N, K, b, res, 1, mean_rows, out=outs[c], cfg=c)
        return ops.gemm_h2(xp, wp, G, M, N, K, b, res, 1, out=outs[c], out_planes=outp.get(c), want_f32=f32o, cfg=c)

    if single:
        for _ in range(single):
            run(cfgs[0])
        torch.cuda.synchronize()
        break
    # bit equality against the first configuration (digest of everything a launch writes)
    def digest(c):
        parts = []
        if outs.get(c) is not None:
            parts.append(outs[c].view(torch.int32).to(torch.int64).sum())
            parts.append((outs[c].view(torch.int32).to(torch.int64) * torch.arange(1, outs[c].numel() + 1, device=dev).view(outs[c].shape) % 1000003).sum())
        if outp.get(c) is not None:
            pv = outp[c].planes.view(torch.int16).to(torch.int64)
            parts.append(pv.sum()); parts.append((pv * (torch.arange(pv.numel(), device=dev) % 8191 + 1)).sum())
            parts.append(outp[c].inv.view(torch.int32).to(torch.int64).sum())
        return tuple(int(p) for p in parts)
    digs = {}
    for c in cfgs:
        if outs.get(c) is not None: outs[c].fill_(float("nan"))
        if outp.get(c) is not None: outp[c].planes.zero_(); outp[c].inv.zero_()
        try:
            run(c); torch.cuda.synchronize()
            digs[c] = digest(c)
        except Exception as e:                            # a configuration this build / shape does not have
            digs[c] = "unsupported (%s)" % (str(e)[:60],)
    live = [c for c in cfgs if not isinstance(digs[c], str)]
    times = {c: [] for c in live}
    for r in range(6):
        for c in live:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6): run(c)
            e1.record(); torch.cuda.synchronize()
            if r: times[c].append(e0.elapsed_time(e1) * 1000 / 6)
    nbytes = 4.0 * G * M * K + 4.0 * (w.numel()) + G * M * N * 4.0 * ((1 if has_res else 0) + (1 if f32o else 0) + (1 if planes else 0))
    for c in cfgs:
        if c not in live:
            print("%-8s %-4d %s" % (name, c, digs[c]), flush=True)
            continue
        med = float(np.median(times[c]))
        print("%-8s %-4d %9.1f %9.1f %8.1f %8.0f  %s" % (name, c, med, min(times[c]), 2.0 * G * M * N * K / med / 1e6, nbytes / med / 1e3,
                                                       "identical" if digs[c] == digs[cfgs[0]] else "DIFFERENT"), flush=True)
    del outs, outp, xp, res
    torch.cuda.empty_cache()
