import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops
dev = torch.device("cuda:0")
np.set_printoptions(linewidth=250, suppress=True)
M, K, N = 256, 128, 128
x = (np.arange(M)[:, None] * 0.0 + np.arange(K)[None, :]).astype(np.float32)
w = np.eye(N, K, dtype=np.float32)
xp, wp = ops.h2_split(torch.from_numpy(x).to(dev)), ops.h2_pack_w(torch.from_numpy(w).to(dev))
y, _ = ops.gemm_h2(xp, wp, 1, M, N, K, cfg=0)
torch.cuda.synchronize()
got = y.cpu().numpy()
print("row 0:", got[0].astype(int))
print("row 37:", got[37].astype(int))
print("rows equal:", bool((got == got[0:1]).all()))
# second experiment: x = one-hot rows over k (x[m][k] = 1 if k == m % 128), W[n][k] = n*128 + k  -> y[m][n] = W[n][m%128]
x2 = np.zeros((M, K), dtype=np.float32); x2[np.arange(M), np.arange(M) % K] = 1.0
w2 = (np.arange(N)[:, None] * 128.0 + np.arange(K)[None, :]).astype(np.float32)
xp2, wp2 = ops.h2_split(torch.from_numpy(x2).to(dev)), ops.h2_pack_w(torch.from_numpy(w2).to(dev))
y2, _ = ops.gemm_h2(xp2, wp2, 1, M, N, K, cfg=0)
torch.cuda.synchronize()
g2 = y2.cpu().numpy()
print("exp2 row 0 (want n*128+0):", g2[0][:24].astype(int)); print("exp2 row 1 (want n*128+1):", g2[1][:24].astype(int)); print("exp2 row 5:", g2[5][:24].astype(int))
print("exp2 col 3 over rows (want 3*128 + m):", g2[:40, 3].astype(int))
