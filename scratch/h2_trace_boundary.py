"""Where do the cycles of a k_gemm_h2 TILE BOUNDARY go?  -DFRCNN_H2_TRACE build: slab stamps (0 top, 1 after the vmcnt wait, 2 after the barrier,
3 after the load burst, 5 after the MFMAs, 6 end incl. fold) of the first 16 workgroups + boundary stamps (0 epilogue start, 1 after
scale / activation, 2 after the float32 stores, 3 after the row-maximum exchange, 4 after the plane stores, 5 epilogue end, 6 after the next
tile's start = residual loads issued).  Prints medians per phase for the slabs right after a boundary against mid-tile slabs.

    python scratch/h2_trace_boundary.py 9,30,31 [shape,...]
"""
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
shapes = {"b3c3x8": (1, 19152, 1024, 256, True, True, True), "b4c3x8": (1, 117600, 2048, 512, True, True, True),
          "b3c3x8f": (1, 19152, 1024, 256, True, True, False), "w7x8": (121, 2400, 512, 512, False, True, False)}
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "9").split(",")]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else ["b3c3x8", "b4c3x8"]
med = lambda a: float(np.median(a)) if len(a) else float("nan")
for name in only:
    G, M, N, K, has_res, f32, planes = shapes[name]
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0)
    w = torch.randn(G, N, K, device=dev) * 0.05
    res = torch.randn(G * M, N, device=dev) if has_res else None
    b = torch.randn(N, device=dev) if G == 1 else None
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    y = torch.empty(G * M, N, device=dev)
    yp = ops.H2.empty(G * M, N, dev) if planes else None
    NSL, NT = 16 * 8 * 64 * 8, 16 * 8 * 32 * 8
    trace = torch.zeros(NSL + NT, dtype=torch.int64, device=dev)
    L.frcnn_h2_set_trace(P(trace.data_ptr()))
    st = P(torch.cuda.current_stream().cuda_stream)
    for cfg in cfgs:
        for rep in range(3):
            trace.zero_()
            rc = L.frcnn_gemm_h2(P(xp.planes.data_ptr()), P(xp.inv.data_ptr()), P(wp[0].data_ptr()), P(wp[1].data_ptr()), None if b is None else P(b.data_ptr()),
                                 None if res is None else P(res.data_ptr()), None, None, P(y.data_ptr()), None if yp is None else P(yp.planes.data_ptr()),
                                 None if yp is None else P(yp.inv.data_ptr()), G, M, N, K, 1, cfg, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
        nw = 4
        tr = trace.cpu().numpy()
        t = tr[:NSL].reshape(16, 8, 64, 8)[:, :nw]                   # [wg][wave][slab][point]
        bt = tr[NSL:].reshape(16, 8, 32, 8)[:, :nw]                  # [wg][wave][tile][point]
        ns = K // 32
        ntile = 64 // ns
        rows = []
        # slabs by position inside their tile (skip tile 0: prologue)
        for pos, label in ((0, "slab 0 after a boundary"), (1, "slab 1"), (2, "slab 2"), (ns - 1, "last slab of a tile")):
            idx = [tl * ns + pos for tl in range(1, ntile)]
            v = t[:, :, idx, :]
            ok = (v[..., 6] > 0) & (v[..., 0] > 0)
            d = lambda a, b_: med((v[..., a] - v[..., b_])[ok])
            rows.append("    %-26s total %6.0f = wait %6.0f + barrier %5.0f + issue %5.0f + mfma %5.0f + fold/end %5.0f" % (label, d(6, 0), d(1, 0), d(2, 1), d(3, 2), d(5, 3), d(6, 5)))
        tl_idx = list(range(1, min(ntile, 31)))
        v = bt[:, :, tl_idx, :]
        last_end = t[:, :, [tl * ns - 1 for tl in tl_idx], 6]       # end of the last slab of tile tl - 1
        first_top = t[:, :, [tl * ns for tl in tl_idx], 0]
        ok = (v[..., 0] > 0) & (v[..., 6] > 0) & (last_end > 0) & (first_top > 0)
        ph = lambda a, b_: med((v[..., a] - v[..., b_])[ok])
        tot = med((first_top - last_end)[ok])
        print("%-8s cfg %d: tile boundary %6.0f cycles = scale/act %5.0f + f32 stores %5.0f + row max / barrier %5.0f + plane stores %5.0f + barrier %5.0f + "
              "step / tile start %5.0f (+ edges %4.0f)" % (name, cfg, tot, ph(1, 0), ph(2, 1), ph(3, 2), ph(4, 3), ph(5, 4), ph(6, 5),
                                                          tot - med((v[..., 6] - v[..., 0])[ok])), flush=True)
        for r in rows:
            print(r, flush=True)
