"""k_gemm_stream vs k_conv_igemm on the path's GEMM shapes: bit-exact check against the automatic configuration + interleaved timing."""
import sys, os, itertools
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
L = lib()
# name: ("conv", M, Cin, Cout, residual) | ("gemm", G, M, N, K)
shapes = {
 "b3c1x4": ("conv", 9576, 1024, 256, False), "b3c3x4": ("conv", 9576, 256, 1024, True),
 "b4c1x4": ("conv", 58800, 2048, 512, False), "b4c3x4": ("conv", 58800, 512, 2048, True),
 "b2c1x4": ("conv", 37500, 512, 128, False), "b2c3x4": ("conv", 37500, 128, 512, True),
 "b1c1x4": ("conv", 150000, 256, 64, False), "b1c3x4": ("conv", 150000, 64, 256, True),
 "w3x4": ("gemm", 36, 640, 256, 256), "w7x4": ("gemm", 121, 1200, 512, 512), "wrpn": ("gemm", 36, 640, 512, 1024),
 "tail": ("conv", 1000, 96, 192, True),
 "b3c1x1": ("conv", 2394, 1024, 256, False), "b3c3x1": ("conv", 2394, 256, 1024, True), "w3x1": ("gemm", 36, 160, 256, 256),
 "b2c1x1": ("conv", 9375, 512, 128, False), "b2c3x1": ("conv", 9375, 128, 512, True),
}
cfgs = [int(c) for c in sys.argv[1].split(",")]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
print("%-7s %4s %9s %9s %8s  %s" % ("shape", "cfg", "med_us", "min_us", "TFLOP/s", "vs auto"))
for name in only:
    sp = shapes[name]
    torch.manual_seed(1)
    if sp[0] == "conv":
        _, M, Cin, Cout, has_res = sp
        x = torch.randn(1, 1, M, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
        res = torch.randn(1, 1, M, Cout, device=dev) if has_res else None
        out = torch.empty(1, 1, M, Cout, device=dev)
        run = lambda: ops.conv2d(x, w, b, 1, 1, 1, (0, 0, 0, 0), 1, res, 1, out=out)
        flops = 2.0 * M * Cin * Cout
    else:
        _, G, M, N, K = sp
        x = torch.randn(G, M, K, device=dev); w = torch.randn(G, N, K, device=dev) * 0.05
        out = torch.empty(G, M, N, device=dev)
        run = lambda: ops.gemm_batched_nt(x, w, out)
        flops = 2.0 * G * M * N * K
    L.frcnn_set_tuning(0, -1); out.fill_(float("nan")); run(); torch.cuda.synchronize(); want = out.clone()
    status = {}
    for cfg in cfgs:
        L.frcnn_set_tuning(0, cfg); out.fill_(float("nan"))
        try:
            run(); torch.cuda.synchronize()
            status[cfg] = "bit-exact" if torch.equal(out, want) else "DIFF max %.3e nan %d" % (float((out - want).abs().nan_to_num(1e30).max()), int(torch.isnan(out).sum()))
        except Exception as e:
            status[cfg] = "unsupported"
    live = [c for c in cfgs if status[c] != "unsupported"]
    times = {c: [] for c in live}
    for r in range(rounds + 1):
        for cfg in live:
            L.frcnn_set_tuning(0, cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run()
            e1.record(); torch.cuda.synchronize()
            if r: times[cfg].append(e0.elapsed_time(e1) * 1000 / 8)
    for cfg in cfgs:
        if cfg in times:
            med = float(np.median(times[cfg]))
            print("%-7s %4d %9.1f %9.1f %8.1f  %s" % (name, cfg, med, min(times[cfg]), flops / med / 1e6, status[cfg]))
        else:
            print("%-7s %4d %9s %9s %8s  %s" % (name, cfg, "-", "-", "-", status[cfg]))
L.frcnn_set_tuning(0, -1)
