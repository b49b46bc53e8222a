"""Toom-Cook / Winograd F(m,3) transform matrices from the evaluation points, in exact rationals, and the block-structured
1-D matrices of the 7-wide special case (7 = 4 + 3).  Prints C initialisers for csrc/winograd7.hip and self-checks.

  y = A^T [ (G g) (.) (B^T d) ],   A^T = V_m^T,  G = D^-1 V_r,  B^T = D V^-T   (V = evaluation matrix incl. the point at infinity)
"""
from fractions import Fraction as F
import numpy as np

def inv(M):
    n = len(M); A = [row[:] + [F(int(i == j)) for j in range(n)] for i, row in enumerate(M)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] != 0); A[c], A[p] = A[p], A[c]
        A[c] = [x / A[c][c] for x in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0: A[r] = [x - A[r][c] * y for x, y in zip(A[r], A[c])]
    return [row[n:] for row in A]

def winograd(points, m, r=3):
    a = [F(p) for p in points]; alpha = m + r - 1; assert len(a) == alpha - 1
    V = [[x ** k for k in range(alpha)] for x in a] + [[F(0)] * (alpha - 1) + [F(1)]]
    Vm = [[x ** k for k in range(m)] for x in a] + [[F(0)] * (m - 1) + [F(1)]]
    Vr = [[x ** k for k in range(r)] for x in a] + [[F(0)] * (r - 1) + [F(1)]]
    N = [np.prod([x - y for y in a if y != x]) for x in a] + [F(1)]
    Vinv = inv(V)
    BT = [[N[j] * Vinv[i][j] for i in range(alpha)] for j in range(alpha)]       # D V^-T
    G = [[Vr[j][k] / N[j] for k in range(r)] for j in range(alpha)]
    AT = [[Vm[j][i] for j in range(alpha)] for i in range(m)]
    return AT, G, BT

def check(AT, G, BT, m):
    rng = np.random.RandomState(0)
    A, Gm, B = (np.array([[float(x) for x in row] for row in M]) for M in (AT, G, BT))
    g, d = rng.randn(3), rng.randn(m + 2)
    y = A @ ((Gm @ g) * (B @ d)); ref = np.array([sum(g[k] * d[i + k] for k in range(3)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-12), (y, ref)

F43 = winograd([0, 1, -1, 2, -2], 4); check(*F43, 4)
F33 = winograd([0, 1, -1, 2], 3); check(*F33, 3)

def cmat(name, M):
    rows = ",\n  ".join("{" + ", ".join(("%.17gf" % float(x)) if x.denominator != 1 else ("%d.f" % x.numerator) for x in row) + "}" for row in M)
    return "static __device__ __constant__ const float %s[%d][%d] = {\n  %s};\n" % (name, len(M), len(M[0]), rows)

# 7-wide: segment A = outputs 0..3 from padded inputs 0..5, segment B = outputs 4..6 from padded inputs 4..8 (padded index = real + 1)
def block(AT4, G4, BT4, AT3, G3, BT3):
    BT7 = [[F(0)] * 9 for _ in range(11)]; AT7 = [[F(0)] * 11 for _ in range(7)]; G7 = [r[:] for r in G4] + [r[:] for r in G3]
    for i in range(6):
        for j in range(6): BT7[i][j] = BT4[i][j]
    for i in range(5):
        for j in range(5): BT7[6 + i][4 + j] = BT3[i][j]
    for i in range(4):
        for j in range(6): AT7[i][j] = AT4[i][j]
    for i in range(3):
        for j in range(5): AT7[4 + i][6 + j] = AT3[i][j]
    return AT7, G7, BT7

AT7, G7, BT7 = block(*F43, *F33)
if __name__ == "__main__":
    rng = np.random.RandomState(1)
    A, Gm, B = (np.array([[float(x) for x in row] for row in M]) for M in (AT7, G7, BT7))
    g = rng.randn(3, 3); d = np.zeros((9, 9)); d[1:8, 1:8] = rng.randn(7, 7)
    Y = A @ ((Gm @ g @ Gm.T) * (B @ d @ B.T)) @ A.T
    ref = np.array([[sum(g[a, b] * d[i + a, j + b] for a in range(3) for b in range(3)) for j in range(7)] for i in range(7)])
    assert np.allclose(Y, ref, atol=1e-11)
    # f32 error level on post-ReLU-like data, 512 channels summed
    C = 512; g32 = (rng.randn(C, 3, 3) * np.sqrt(2.0 / (9 * C))).astype(np.float32); x = np.maximum(rng.randn(C, 7, 7), 0).astype(np.float32)
    dp = np.zeros((C, 9, 9), np.float32); dp[:, 1:8, 1:8] = x
    A32, G32, B32 = A.astype(np.float32), Gm.astype(np.float32), B.astype(np.float32)
    U = np.einsum("ia,cab,jb->cij", Gm, g32.astype(np.float64), Gm).astype(np.float32)
    V = np.einsum("ia,cab,jb->cij", B32, dp, B32)
    Msum = (U * V).sum(0, dtype=np.float32)
    Y32 = A32 @ Msum @ A32.T
    ref = sum(np.array([[np.sum(g32[c].astype(np.float64) * dp[c, i:i + 3, j:j + 3]) for j in range(7)] for i in range(7)]) for c in range(C))
    print("7x7 mixed F(4,3)+F(3,3): 121 products per channel pair (F(4,3) alone: 144, direct: 441); f32 rel err %.2e" % (np.abs(Y32 - ref).max() / np.abs(ref).max()))
    print(cmat("W7_BT", BT7)); print(cmat("W7_AT", AT7)); print(cmat("W7_G", G7))
