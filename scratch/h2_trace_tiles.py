"""What does a TILE BOUNDARY of k_gemm_h2 cost?  -DFRCNN_H2_TRACE stamps every slab of the first 16 workgroups (0 = slab top ... 6 = slab
end); the gap between a slab's end and the next slab's top is the loop edge inside a tile and epilogue + init_tot + tile stepping at a
tile boundary.  Short-K shapes (block3 conv3: 8 slabs per tile, residual, f32 + planes out) with several tiles per workgroup."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
csrc = os.path.join(ROOT, "tf-faster-rcnn_amd", "csrc")
so = "/tmp/libh2trace.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + csrc, "-DFRCNN_H2_TRACE", "-shared", os.path.join(csrc, "gemm_h2.hip"), "-o", so])
L = ctypes.CDLL(so)
from frcnn_hip import ops
dev = torch.device("cuda:0")
P = ctypes.c_void_p
shapes = {"b3c3x12 res f32+planes": (1, 28728, 1024, 256, True, True, True), "b3c3x12 res f32": (1, 28728, 1024, 256, True, True, False),
          "b3c3x12 plain f32": (1, 28728, 1024, 256, False, True, False), "b4c3x4 res f32+planes": (1, 58800, 2048, 512, True, True, True),
          "w7x4": (121, 1200, 512, 512, False, True, False)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for name, (G, M, N, K, has_res, f32, planes) in shapes.items():
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0)
    w = torch.randn(G, N, K, device=dev) * 0.05
    res = torch.randn(G * M, N, device=dev) if has_res else None
    b = torch.randn(N, device=dev) if G == 1 else None
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    y = torch.empty(G * M, N, device=dev)
    yp = ops.H2.empty(G * M, N, dev) if planes else None
    trace = torch.zeros(16 * 8 * 64 * 8, dtype=torch.int64, device=dev)
    L.frcnn_h2_set_trace(P(trace.data_ptr()))
    st = P(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        trace.zero_()
        rc = L.frcnn_gemm_h2(P(xp.planes.data_ptr()), P(xp.inv.data_ptr()), P(wp[0].data_ptr()), P(wp[1].data_ptr()), None if b is None else P(b.data_ptr()),
                             None if res is None else P(res.data_ptr()), None, None, P(y.data_ptr()), None if yp is None else P(yp.planes.data_ptr()),
                             None if yp is None else P(yp.inv.data_ptr()), G, M, N, K, 1, cfg, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
    nw = 8 if cfg == 21 else 4
    t = trace.cpu().numpy().reshape(16, 8, 64, 8)[:, :nw]
    ns = K // 32
    ok = (t[:, :, 1:, 0] > 0) & (t[:, :, :-1, 6] > 0)
    gap = t[:, :, 1:, 0] - t[:, :, :-1, 6]                              # slab s end -> slab s + 1 top
    idx = np.arange(63)
    edge = (idx % ns) == ns - 1
    slab = (t[..., 6] - t[..., 0])[t[..., 6] > 0]
    ge, gi = gap[:, :, edge][ok[:, :, edge]], gap[:, :, ~edge][ok[:, :, ~edge]]
    print("%-26s cfg %d: slab %6.0f cycles (median), %d slabs per tile; gap inside a tile %5.0f, gap at a TILE BOUNDARY median %6.0f  p10 %6.0f  p90 %6.0f  (= %.1f slabs)"
          % (name, cfg, np.median(slab), ns, np.median(gi), np.median(ge), np.percentile(ge, 10), np.percentile(ge, 90), np.median(ge) / np.median(slab)), flush=True)
