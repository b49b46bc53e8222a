"""Is k_gemm_h2 bound by the L2 -> L1 -> LDS path rather than by its instruction schedule?  A 32-k slab row of one operand plane is
64 bytes = HALF a 128-byte cache line, so every direct-to-LDS instruction (16 rows x 64 B) touches 16 lines and uses half of each; the
other half is the NEXT slab's data, which the 32 KB L1 has dropped by then.  cfg 20 (-DFRCNN_ABLATION build; wrong results by
construction) moves the same bytes per slab with the same instruction count, but as 8 rows x one full line per instruction.
A/B in one process, interleaved, on the big GEMM shapes of the path."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
csrc = os.path.join(ROOT, "tf-faster-rcnn_amd", "csrc")
so = "/tmp/libh2abl.so"
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + csrc, "-DFRCNN_ABLATION", "-shared", os.path.join(csrc, "gemm_h2.hip"), "-o", so])
L = ctypes.CDLL(so)
from frcnn_hip import ops
dev = torch.device("cuda:0")
P = ctypes.c_void_p
shapes = {"b4c1x4": (1, 58800, 512, 2048), "b4c3x4": (1, 58800, 2048, 512), "w7x4": (121, 1200, 512, 512), "b3c1x4": (1, 9576, 256, 1024),
          "b3c3x4": (1, 9576, 1024, 256), "b3c1x12": (1, 28728, 256, 1024)}
if len(sys.argv) > 2: shapes = {k: v for k, v in shapes.items() if k in sys.argv[2].split(",")}
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "9,20").split(",")]
for name, (G, M, N, K) in shapes.items():
    torch.manual_seed(1)
    x = torch.randn(G * M, K, device=dev).clamp(min=0)
    w = torch.randn(G, N, K, device=dev) * 0.05
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    y = torch.empty(G * M, N, device=dev)
    st = P(torch.cuda.current_stream().cuda_stream)
    def run(c):
        rc = L.frcnn_gemm_h2(P(xp.planes.data_ptr()), P(xp.inv.data_ptr()), P(wp[0].data_ptr()), P(wp[1].data_ptr()), None, None, None, None,
                             P(y.data_ptr()), None, None, G, M, N, K, 1, c, st)
        assert rc == 0, rc
    times = {c: [] for c in cfgs}
    for r in range(6):
        for c in cfgs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run(c)
            e1.record(); torch.cuda.synchronize()
            if r: times[c].append(e0.elapsed_time(e1) * 1000 / 8)
    for c in cfgs:
        med = float(np.median(times[c]))
        print("%-8s cfg %2d  %8.1f us  %7.1f TFLOP/s f32-eq" % (name, c, med, 2.0 * G * M * N * K / med / 1e6), flush=True)
