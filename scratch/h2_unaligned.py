"""Does global_load_lds_dwordx4 need a 16-byte aligned global address?  frcnn_gemm_h2 fetches the block scales x_inv with it; pass an
x_inv whose base is only 4-byte aligned and compare with the aligned run (decides whether the M % 4 restriction can go)."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import torch
from frcnn_hip import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, N = 128 * 9 + 4, 512, 256
x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev) * 3)
w = torch.randn(N, K, device=dev) * 0.05
xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
y0, _ = ops.gemm_h2(xp, wp, 1, M, N, K)
for off in (1, 2, 3):
    store = torch.zeros(xp.inv.numel() + 8, dtype=torch.float32, device=dev)
    inv2 = store[off:off + xp.inv.numel()].view_as(xp.inv)
    inv2.copy_(xp.inv)
    xq = ops.H2(xp.planes, inv2, xp.rows, xp.K)
    y1, _ = ops.gemm_h2(xq, wp, 1, M, N, K)
    torch.cuda.synchronize()
    print("x_inv base offset %d floats (address %% 16 = %d): identical = %s, max |diff| = %.3g" % (off, inv2.data_ptr() % 16, bool(torch.equal(y0, y1)), float((y0 - y1).abs().max())))
