"""Where do the cycles of k_gemm_h2 go?  Builds csrc/gemm_h2.hip with -DFRCNN_H2_TRACE (s_memtime stamps at six points of every slab of
the first 16 workgroups: 0 loop top, 1 after the vmcnt wait, 2 after the barrier, 3 after the direct-to-LDS issue burst, 4 between the
two 16-k MFMA groups, 5 after the last MFMA was issued, 6 after the fold) into a scratch library and prints per-segment medians."""
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
shapes = {"b4c1x4": (1, 58800, 512, 2048), "b3c3x4": (1, 9576, 1024, 256), "w7x4": (121, 1200, 512, 512)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 9
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
    t = trace.cpu().numpy().reshape(16, 8, 64, 8)[:, :4]             # [wg][wave][slab][point]
    ok = t[..., 6] > 0
    seg = {"wait vmcnt (0->1)": t[..., 1] - t[..., 0], "barrier (1->2)": t[..., 2] - t[..., 1], "DMA issue burst (2->3)": t[..., 3] - t[..., 2],
           "reads + MFMA group 0 (3->4)": t[..., 4] - t[..., 3], "MFMA group 1 (4->5)": t[..., 5] - t[..., 4], "advance + fold (5->6)": t[..., 6] - t[..., 5],
           "whole slab (0->6)": t[..., 6] - t[..., 0]}
    sl = slice(8, 56)                                                 # steady state
    print("%s cfg %d (s_memtime ticks = shader cycles; medians over 16 workgroups x 4 waves x slabs 8..55)" % (name, cfg))
    for k_, v in seg.items():
        vv = v[:, :, sl][ok[:, :, sl]]
        print("  %-30s median %6.0f   p10 %6.0f   p90 %6.0f" % (k_, np.median(vv), np.percentile(vv, 10), np.percentile(vv, 90)))
    nxt = t[:, :, 1:, 0] - t[:, :, :-1, 6]
    print("  %-30s median %6.0f" % ("loop back edge (6->next 0)", np.median(nxt[:, :, sl][ok[:, :, 1:][:, :, sl]])))
    print("  MFMA issue needs 24 x 32 = 768 cycles per slab per wave; two waves share a SIMD", flush=True)
