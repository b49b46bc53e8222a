"""A/B harness: configurations interleaved in rounds inside one process, median over rounds."""
import sys, os, itertools
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
L = lib()
shapes = {  # name: (N,H,W,Cin,Cout,k,pad)
 "b3c1": (1,38,63,1024,256,1,0), "b3c2": (1,38,63,256,256,3,1), "b3c3": (1,38,63,256,1024,1,0),
 "rpn":  (1,38,63,1024,512,3,1), "b2c2": (1,75,125,128,128,3,1), "b2c3": (1,75,125,128,512,1,0), "b2c1": (1,75,125,512,128,1,0),
 "b1c2": (1,150,250,64,64,3,1), "b1c3": (1,150,250,64,256,1,0),
 "b3c1x4": (4,38,63,1024,256,1,0), "b3c2x4": (4,38,63,256,256,3,1), "b3c3x4": (4,38,63,256,1024,1,0), "rpnx4": (4,38,63,1024,512,3,1),
 "b2c2x4": (4,75,125,128,128,3,1), "b2c3x4": (4,75,125,128,512,1,0), "b2c1x4": (4,75,125,512,128,1,0), "b1c2x4": (4,150,250,64,64,3,1),
 "b1c3x4": (4,150,250,64,256,1,0), "b1c1x4": (4,150,250,256,64,1,0),
 "b4c1x4": (1200,7,7,2048,512,1,0), "b4c3x4": (1200,7,7,512,2048,1,0),
 "b4c1": (300,7,7,2048,512,1,0), "b4c2": (300,7,7,512,512,3,1), "b4c3": (300,7,7,512,2048,1,0),
}
cfgs = [int(c) for c in sys.argv[1].split(",")]
dbgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(shapes)
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
print("%-6s %4s %3s %9s %9s %8s" % ("shape", "cfg", "dbg", "med_us", "min_us", "TFLOP/s"))
for name in only:
    N,H,W,Cin,Cout,k,pad = shapes[name]
    x = torch.randn(N,H,W,Cin, device=dev); w = torch.randn(Cout,k,k,Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    res = torch.randn(N,H,W,Cout, device=dev) if name.endswith("c3") else None
    out = torch.empty(N,H,W,Cout, device=dev)
    flops = 2.0*N*H*W*Cout*k*k*Cin
    combos = list(itertools.product(cfgs, dbgs))
    times = {c: [] for c in combos}
    for r in range(rounds + 1):
        for cfg, dbg in combos:
            if cfg >= 100: L.frcnn_set_tuning(2, 1); L.frcnn_set_tuning(3, cfg - 100)
            else: L.frcnn_set_tuning(2, 0); L.frcnn_set_tuning(0, cfg)
            L.frcnn_set_tuning(1, dbg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.conv2d(x, w, b, k, k, 1, (pad,)*4, 1, res, 1, out=out)
            e1.record(); torch.cuda.synchronize()
            if r: times[(cfg, dbg)].append(e0.elapsed_time(e1) * 1000 / 8)
    for (cfg, dbg), ts in times.items():
        med = float(np.median(ts))
        print("%-6s %4d %3d %9.1f %9.1f %8.1f" % (name, cfg, dbg, med, min(ts), flops / med / 1e6))
L.frcnn_set_tuning(0, -1); L.frcnn_set_tuning(1, 0); L.frcnn_set_tuning(2, 0); L.frcnn_set_tuning(3, -1)
