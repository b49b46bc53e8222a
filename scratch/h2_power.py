"""Clock and socket power while ONE frcnn_gemm_h2 configuration runs back to back for ~2.5 s (amdgpu sysfs nodes, bench.py's sampler):
is a schedule that keeps the matrix pipe busier paid back in time, or taken away again by the power limit?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd"), ROOT]
import numpy as np, torch
from frcnn_hip import ops
import bench
dev = torch.device("cuda:0")
shapes = {"b4c1x4": (1, 58800, 512, 2048, False), "b4c3x4": (1, 58800, 2048, 512, True), "w7x4": (121, 1200, 512, 512, False),
          "b3c1x12": (1, 28728, 256, 1024, False)}
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "9,21").split(",")]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5
tel = bench.Telemetry()
print("%-8s %-4s %9s %8s %7s %7s %9s" % ("shape", "cfg", "us", "TFLOP/s", "MHz", "W", "J/TFLOP"))
for name in only:
    G, M, N, K, has_res = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 4 - 2)
    w = torch.randn(G, N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if G == 1 else None
    res = torch.randn(G * M, N, device=dev) if has_res else None
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    y = torch.empty(G * M, N, device=dev)
    for c in cfgs:
        out = {}
        def region():
            run = lambda: ops.gemm_h2(xp, wp, G, M, N, K, b, res, 1, out=y, cfg=c)
            for _ in range(20): run()
            torch.cuda.synchronize()
            n, t0 = 0, time.time()
            while time.time() - t0 < secs:
                for _ in range(50): run()
                torch.cuda.synchronize()
                n += 50
            out["us"] = (time.time() - t0) / n * 1e6
        t = tel.run(region)
        tf = 2.0 * G * M * N * K / out["us"] / 1e6
        mhz, wt = (t or {}).get("sclk_mhz"), (t or {}).get("socket_w")
        print("%-8s %-4d %9.1f %8.1f %7s %7s %9s" % (name, c, out["us"], tf, mhz, wt, "%.2f" % (wt / tf) if wt else None), flush=True)
        time.sleep(0.5)
