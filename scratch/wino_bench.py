"""The Winograd transforms on the shapes of a bench step (8 images per launch): time, algorithmic GB/s, and the row-per-thread form
against the tile-per-thread form (frcnn_set_tuning(9, n): the workgroup count below which the F(4,3) transforms run row-per-thread),
bit equality of everything written.  Round 6 (the energy ledger puts the transforms at 11 % of the step's joules).

    python scratch/wino_bench.py [thresholds, default 256,100000000]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
import numpy as np, torch
import frcnn_hip
from frcnn_hip import ops
dev = torch.device("cuda:0")
ths = [int(a) for a in (sys.argv[1] if (len(sys.argv) > 1 and "=" not in sys.argv[1]) else "256,100000000").split(",")]
shapes = {  # name: (N, H, W, C, m)
    "block3 conv2 x8": (8, 38, 63, 256, 4),
    "block3 conv2 x1": (1, 38, 63, 256, 4),
    "rpn 3x3 x8": (8, 38, 63, 1024, 4),
    "block2 conv2 x8 (F2)": (8, 75, 125, 128, 2),
    "tail conv2 7x7 x8": (2400, 7, 7, 512, 7),
}
single = [a for a in sys.argv[1:] if a.startswith("single=")]      # single=<shape name>:<input|output>:<launches>  (rocprofv3 --pmc passes: one kernel, nothing else)
if single:
    nm, which, n = single[0][7:].split(":")
    N, H, W, C, m = shapes[nm]
    x = torch.randn(N, H, W, C, device=dev).clamp(min=0)
    G, T = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
    mm = torch.randn(G, T, C, device=dev)
    bias = torch.randn(C, device=dev)
    v, yp = ops.H2.empty(G * T, C, dev), ops.H2.empty(N * H * W, C, dev)
    for _ in range(int(n)):
        if which == "input":
            ops.winograd_input_transform_h2(x, v, m)
        else:
            ops.winograd_output_transform_h2(mm, bias, 1, (N, H, W, C), m, yp)
    torch.cuda.synchronize()
    print("algorithmic bytes per launch: %d" % (4 * (N * H * W * C + G * T * C)))
    sys.exit(0)
print("%-24s %-10s %-6s %9s %9s %8s  %s" % ("shape", "transform", "rows<", "med_us", "min_us", "GB/s", "bits vs first"))
for name, (N, H, W, C, m) in shapes.items():
    torch.manual_seed(1)
    x = torch.randn(N, H, W, C, device=dev).clamp(min=0)
    G, T = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
    mm = torch.randn(G, T, C, device=dev)
    bias = torch.randn(C, device=dev)
    for which in ("input", "output"):
        ref = None
        for th in ths:
            frcnn_hip.lib().frcnn_set_tuning(9, th)
            v = ops.H2.empty(G * T, C, dev)
            yp = ops.H2.empty(N * H * W, C, dev)
            v.planes.zero_(); v.inv.zero_(); yp.planes.zero_(); yp.inv.zero_()

            def run():
                if which == "input":
                    ops.winograd_input_transform_h2(x, v, m)
                else:
                    ops.winograd_output_transform_h2(mm, bias, 1, (N, H, W, C), m, yp)
            run(); torch.cuda.synchronize()
            o = v if which == "input" else yp
            pv = o.planes.view(torch.int16).to(torch.int64).view(-1)
            dig = (int(pv.sum()), int((pv * (torch.arange(pv.numel(), device=dev) % 8191 + 1)).sum()), int(o.inv.view(torch.int32).to(torch.int64).sum()))
            del pv
            ref = dig if ref is None else ref
            ts = []
            for r in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize()
                if r: ts.append(e0.elapsed_time(e1) * 100)
            nbytes = 4.0 * (N * H * W * C + G * T * C)
            print("%-24s %-10s %-6s %9.1f %9.1f %8.0f  %s" % (name, which, th if th < 10**8 else "all", float(np.median(ts)), min(ts), nbytes / np.median(ts) / 1e3,
                                                             "identical" if dig == ref else "DIFFERENT"), flush=True)
            if m != 4:
                break
frcnn_hip.lib().frcnn_set_tuning(9, 0)
