"""s_memtime stamps inside the ping-pong schedule of k_gemm_h2 (cfg 21; -DFRCNN_H2_TRACE build): per wave and slab, 0 MEM start, 1 after
the 16 fragment reads + the slab q + 2 loads were issued (+ fold), 2 after the waits (group 1: vmcnt; both: lgkmcnt(0)), 3 after the
barrier = MFMA start, 4 after the 24 MFMAs were issued, 5 after group 0's vmcnt wait, 6 after the closing barrier."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
csrc = os.path.join(ROOT, "tf-faster-rcnn_amd", "csrc")
so = "/tmp/libh2trace.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + csrc, "-DFRCNN_H2_TRACE", "-DFRCNN_ABLATION", "-shared", os.path.join(csrc, "gemm_h2.hip"), "-o", so])
L = ctypes.CDLL(so)
from frcnn_hip import ops
dev = torch.device("cuda:0")
P = ctypes.c_void_p
shapes = {"b4c1x4": (1, 58800, 512, 2048), "b4c3x4": (1, 58800, 2048, 512), "w7x4": (121, 1200, 512, 512)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
for name, (G, M, N, K) in shapes.items():
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0)
    w = torch.randn(G, N, K, device=dev) * 0.05
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    y = torch.empty(G * M, N, device=dev)
    trace = torch.zeros(16 * 8 * 64 * 8, dtype=torch.int64, device=dev)
    L.frcnn_h2_set_trace(P(trace.data_ptr()))
    st = P(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        trace.zero_()
        rc = L.frcnn_gemm_h2(P(xp.planes.data_ptr()), P(xp.inv.data_ptr()), P(wp[0].data_ptr()), P(wp[1].data_ptr()), None, None, None, None,
                             P(y.data_ptr()), None, None, G, M, N, K, 1, cfg, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(16, 8, 64, 8)                     # [wg][wave][slab][point]
    sl = slice(6, 60)
    for grp, name_g in ((0, "group 0 (waves 0-3)"), (1, "group 1 (waves 4-7)")):
        tt = t[:, 4 * grp:4 * grp + 4, sl]
        ok = tt[..., 6] > 0
        seg = {"MEM: reads + issue + fold (0->1)": tt[..., 1] - tt[..., 0], "MEM: waits (1->2)": tt[..., 2] - tt[..., 1], "barrier (2->3)": tt[..., 3] - tt[..., 2],
               "MFMA issue (3->4)": tt[..., 4] - tt[..., 3], "MFMA-end vmcnt wait (4->5)": tt[..., 5] - tt[..., 4], "barrier (5->6)": tt[..., 6] - tt[..., 5],
               "whole slab (0->6)": tt[..., 6] - tt[..., 0]}
        print("%s cfg %d %s (shader cycles; medians over 16 workgroups x 4 waves x slabs 6..59)" % (name, cfg, name_g))
        for k_, v in seg.items():
            vv = v[ok]
            print("  %-36s median %6.0f   p10 %6.0f   p90 %6.0f" % (k_, np.median(vv), np.percentile(vv, 10), np.percentile(vv, 90)))
    print("  one slab = 24 MFMAs = 768 matrix-pipe cycles per wave; two waves (one per group) share a SIMD", flush=True)
