"""frcnn_gemm_h2: the one-barrier-per-slab schedule (cfg 9 = shipped in round 3, cfg 3 = its 256 x 128 / 8-wave form) against the
ping-pong schedule (cfg 21) on the GEMM shapes of the path: time (interleaved A/B in one process), f32-equivalent TFLOP/s, bit equality."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
import ablation_lib; ablation_lib.use()
import numpy as np, torch
from frcnn_hip import ops
dev = torch.device("cuda:0")
shapes = {  # name: (G, M, N, K, residual, planes out)
 "b4c1x4": (1, 58800, 512, 2048, False, True), "b4c3x4": (1, 58800, 2048, 512, True, True), "b4c3x4f": (1, 58800, 2048, 512, True, False),
 "w7x4": (121, 1200, 512, 512, False, False), "wrpn": (36, 640, 512, 1024, False, False),
 "b3c1x4": (1, 9576, 256, 1024, False, True), "b3c3x4": (1, 9576, 1024, 256, True, True), "w3x4": (36, 640, 256, 256, False, False),
 "b3c1x12": (1, 28728, 256, 1024, False, True), "b3c3x12": (1, 28728, 1024, 256, True, True),
 "b4c1x1": (1, 14700, 512, 2048, False, True), "b4c3x1": (1, 14700, 2048, 512, True, True),
 "b3c1x1": (1, 2394, 256, 1024, False, True), "b3c3x1": (1, 2394, 1024, 256, True, True), "w3x1": (36, 160, 256, 256, False, False),
 "b2c1x4": (1, 37500, 128, 512, False, True), "b2c3x4": (1, 37500, 512, 128, True, True), "b3scx4": (1, 9576, 1024, 512, False, True),
}
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "9,3,21").split(",")]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
print("%-8s %-4s %9s %9s %8s  %s" % ("shape", "cfg", "med_us", "min_us", "TFLOP/s", "bits vs cfg %d" % cfgs[0]))
for name in only:
    G, M, N, K, has_res, planes = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 4 - 2)
    w = torch.randn(G, N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if G == 1 else None
    res = torch.randn(G * M, N, device=dev) if has_res else None
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    outs, outp = {}, {}
    for c in cfgs:
        outs[c] = torch.empty(G * M, N, device=dev)
        outp[c] = ops.H2.empty(G * M, N, dev) if planes else None
    run = lambda c: ops.gemm_h2(xp, wp, G, M, N, K, b, res, 1, out=outs[c], out_planes=outp[c], cfg=c)
    times = {c: [] for c in cfgs}
    for r in range(6):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run(c)
            e1.record(); torch.cuda.synchronize()
            if r: times[c].append(e0.elapsed_time(e1) * 1000 / 8)
    for c in cfgs:
        same = torch.equal(outs[c].view(torch.int32), outs[cfgs[0]].view(torch.int32)) and (not planes or torch.equal(outp[c].planes, outp[cfgs[0]].planes))
        med = float(np.median(times[c]))
        print("%-8s %-4d %9.1f %9.1f %8.1f  %s" % (name, c, med, min(times[c]), 2.0 * G * M * N * K / med / 1e6, "identical" if same else "DIFFERENT"), flush=True)
