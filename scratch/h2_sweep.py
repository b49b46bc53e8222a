"""frcnn_gemm_h2 (block-scaled fp16x2 operands, 3 MFMAs / product) vs frcnn_gemm_x3 (bf16x3, 6 MFMAs) vs the f32-MFMA kernels:
error against float64 and time, on the GEMM shapes of the ResNet-101 path (4-image batches)."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
L = lib()
shapes = {  # name: (G, M, N, K, residual, act)
 "b4c1x4": (1, 58800, 512, 2048, False, 1), "b4c3x4": (1, 58800, 2048, 512, True, 1), "w7x4": (121, 1200, 512, 512, False, 0),
 "wrpn": (36, 640, 512, 1024, False, 0), "b3c1x4": (1, 9576, 256, 1024, False, 1), "b3c3x4": (1, 9576, 1024, 256, True, 1),
 "w3x4": (36, 640, 256, 256, False, 0), "b2c3x4": (1, 37500, 512, 128, True, 1), "b2c1x4": (1, 37500, 128, 512, False, 1),
 "b4c1x1": (1, 14700, 512, 2048, False, 1), "b4c3x1": (1, 14700, 2048, 512, True, 1), "w7x1": (121, 300, 512, 512, False, 0),
 "b3c1x1": (1, 2396, 256, 1024, False, 1), "b3c3x1": (1, 2396, 1024, 256, True, 1),
}
cfgs = [int(c) for c in sys.argv[1].split(",")]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
planes_out = len(sys.argv) > 3 and sys.argv[3] == "planes"
rounds = 5
print("%-7s %-8s %9s %9s %8s  %s" % ("shape", "kernel", "med_us", "min_us", "TFLOP/s", "max err vs f64 / scale"))
for name in only:
    G, M, N, K, has_res, act = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 4 - 2)
    w = torch.randn(G, N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if G == 1 else None
    res = torch.randn(G * M, N, device=dev) if has_res else None
    flops = 2.0 * G * M * N * K
    rows = torch.arange(0, M, max(1, M // 256), device=dev)[:256]
    xr = x.view(G, M, K)[:, rows]
    ref = torch.einsum("gmk,gnk->gmn", xr.double(), w.double())
    if b is not None: ref = ref + b.double()
    if res is not None: ref = ref + res.view(G, M, N)[:, rows].double()
    if act == 1: ref = ref.clamp(min=0)
    if act == 2: ref = ref.clamp(min=0, max=6)
    scale = float(ref.abs().max())
    err_of = lambda t: float((t.view(G, M, N)[:, rows].double() - ref).abs().max()) / scale
    runs, outs = {}, {}
    out32 = torch.empty(G * M, N, device=dev)
    if G == 1:
        runs["f32"] = lambda: ops.conv2d(x.view(1, 1, M, K), w.view(N, 1, 1, K), b, 1, 1, 1, (0, 0, 0, 0), act, None if res is None else res.view(1, 1, M, N), 1, out=out32.view(1, 1, M, N))
    elif res is None and act == 0:
        runs["f32"] = lambda: ops.gemm_batched_nt(x.view(G, M, K), w, out32.view(G, M, N))
    outs["f32"] = out32
    x3p = ops.gemm_x3_pack(w)
    out3 = torch.empty(G * M, N, device=dev)
    runs["x3"] = lambda: ops.gemm_x3(x.view(G, M, K) if G > 1 else x, x3p, G, M, N, K, b, res, act, out=out3.view(G, M, N) if G > 1 else out3)
    outs["x3"] = out3
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    yp = ops.H2.empty(G * M, N, dev) if planes_out else None
    ypp = ops.H2.empty(G * M, N, dev)
    for c in cfgs:
        o = torch.empty(G * M, N, device=dev)
        outs["h2/%d" % c] = o
        runs["h2/%d" % c] = (lambda c, o: (lambda: ops.gemm_h2(xp, wp, G, M, N, K, b, res, act, out=o, out_planes=yp, cfg=c)))(c, o)
        if c in (0, 12) and not planes_out:          # A/B: the same launch also emitting the result's operand planes / planes only
            runs["h2/%d+p" % c] = (lambda c, o: (lambda: ops.gemm_h2(xp, wp, G, M, N, K, b, res, act, out=o, out_planes=ypp, cfg=c)))(c, o)
            runs["h2/%d=p" % c] = (lambda c: (lambda: ops.gemm_h2(xp, wp, G, M, N, K, b, res, act, out=None, out_planes=ypp, want_f32=False, cfg=c)))(c)
    xs = ops.H2.empty(G * M, K, dev)
    runs["split"] = lambda: ops.h2_split(x, out=xs)
    errs = {}
    for k_, f in runs.items():
        if k_ == "split" or k_ not in outs: continue
        outs[k_].fill_(float("nan")); f(); torch.cuda.synchronize()
        errs[k_] = "%.2e nan %d" % (err_of(outs[k_]), int(torch.isnan(outs[k_]).sum()))
    times = {k_: [] for k_ in runs}
    for r in range(rounds + 1):
        for k_, f in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): f()
            e1.record(); torch.cuda.synchronize()
            if r: times[k_].append(e0.elapsed_time(e1) * 1000 / 8)
    for k_ in runs:
        med = float(np.median(times[k_]))
        print("%-7s %-8s %9.1f %9.1f %8.1f  %s" % (name, k_, med, min(times[k_]), (flops / med / 1e6) if k_ != "split" else 0.0, errs.get(k_, "")), flush=True)
