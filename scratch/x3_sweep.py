"""frcnn_gemm_x3 (exact bf16x3 split on the bf16 matrix pipe) vs the f32-MFMA kernels: error against float64 and time."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
L = lib()
shapes = {  # name: (G, M, N, K, residual, act)
 "b4c1x4": (1, 58800, 512, 2048, False, 1), "b4c3x4": (1, 58800, 2048, 512, True, 1), "w7x4": (121, 1200, 512, 512, False, 0),
 "wrpn": (36, 640, 512, 1024, False, 0), "b3c1x4": (1, 9576, 256, 1024, False, 1), "b3c3x4": (1, 9576, 1024, 256, True, 1),
 "w3x4": (36, 640, 256, 256, False, 0), "small": (3, 333, 128, 96, True, 2),
 "b1c1x4": (1, 150000, 64, 256, False, 1), "w1x4": (16, 37500, 64, 64, False, 0), "b1c3x4": (1, 150000, 256, 64, True, 1),
 "b3c1x1": (1, 2394, 256, 1024, False, 1), "b3c3x1": (1, 2394, 1024, 256, True, 1), "w3x1": (36, 160, 256, 256, False, 0), "b4c1x1": (1, 14700, 512, 2048, False, 1),
 "b4c3x1": (1, 14700, 2048, 512, True, 1), "w7x1": (121, 300, 512, 512, False, 0), "b2c3x4": (1, 37500, 512, 128, True, 1), "b2c1x4": (1, 37500, 128, 512, False, 1),
}
cfgs = [int(c) for c in sys.argv[1].split(",")]
X3_TERMS = int(os.environ.get("X3_TERMS", "6"))
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
rounds = 5
print("%-7s %-6s %9s %9s %8s  %s" % ("shape", "kernel", "med_us", "min_us", "TFLOP/s", "max err vs f64 / scale   (f32 kernel err)"))
for name in only:
    G, M, N, K, has_res, act = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G, M, K, device=dev); w = torch.randn(G, N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if G == 1 else None
    res = torch.randn(G, M, N, device=dev) if has_res else None
    flops = 2.0 * G * M * N * K
    # float64 reference on a sample of rows (full reference for the small shapes)
    rows = torch.arange(0, M, max(1, M // 256), device=dev)[:256]
    ref = torch.einsum("gmk,gnk->gmn", x[:, rows].double(), w.double())
    if b is not None: ref = ref + b.double()
    if res is not None: ref = ref + res[:, rows].double()
    if act == 1: ref = ref.clamp(min=0)
    if act == 2: ref = ref.clamp(min=0, max=6)
    scale = float(ref.abs().max())
    out32 = torch.empty(G, M, N, device=dev)
    if G == 1:
        run32 = lambda: ops.conv2d(x.view(1, 1, M, K), w.view(N, 1, 1, K), b, 1, 1, 1, (0, 0, 0, 0), act, None if res is None else res.view(1, 1, M, N), 1, out=out32.view(1, 1, M, N))
    else:
        run32 = (lambda: ops.gemm_batched_nt(x, w, out32)) if (res is None and act == 0) else None
    err32 = None
    if run32 is not None:
        run32(); torch.cuda.synchronize()
        err32 = float((out32[:, rows].double() - ref).abs().max()) / scale
    planes = ops.gemm_x3_pack(w)
    out = torch.full((G, M + 8, N), 7.25, device=dev)
    outv = out[:, :M] if G == 1 else None
    runs = {}
    if run32 is not None: runs["f32"] = run32
    for c in cfgs:
        def mk(c):
            def f():
                if G == 1:
                    ops.gemm_x3(x, planes, 1, M, N, K, b, res, act, out=out[0, :M], cfg=c, terms=X3_TERMS)
                else:
                    ops.gemm_x3(x, planes, G, M, N, K, b, res, act, out=y3, cfg=c, terms=X3_TERMS)
            return f
        runs["x3/%d" % c] = mk(c)
    y3 = torch.empty(G, M, N, device=dev)
    errs = {}
    for k_, f in runs.items():
        if k_ == "f32": continue
        y3.fill_(float("nan")); out.fill_(7.25)
        f(); torch.cuda.synchronize()
        got = out[:, :M] if G == 1 else y3
        errs[k_] = "%.2e nan %d guard %s" % (float((got[:, rows].double() - ref).abs().max()) / scale, int(torch.isnan(got).sum()),
                                              "ok" if (G > 1 or bool((out[:, M:] == 7.25).all())) else "OVERRUN")
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
        print("%-7s %-6s %9.1f %9.1f %8.1f  %s" % (name, k_, med, min(times[k_]), flops / med / 1e6,
                                                    ("%.2e" % err32) if k_ == "f32" else errs[k_]))
