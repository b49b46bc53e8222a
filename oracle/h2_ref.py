"""TEST INFRASTRUCTURE ONLY -- numpy statement of the operand format of csrc/gemm_h2.hip (cfg.HIP.MFMA_H2): a float32 value of a
128-k block = (h + l) * 2^-e with h, l fp16 (both rounded to nearest even) and ONE exact power-of-two scale per (row, block); the
weights use one scale per output row.  Used only by tests/: the device kernels (frcnn_h2_split, frcnn_h2_pack_w, the plane-emitting
epilogue of frcnn_gemm_h2, the plane-emitting Winograd transforms with their row-group scales) must reproduce these arrays bit for bit."""
import numpy as np

KB = 128


def block_scale(mx):
    """(2^e, 2^-e) with mx * 2^e in [2^14, 2^15): exponent arithmetic on the float32 bits of mx >= 0 (h2_block_scale)."""
    mx = np.asarray(mx, dtype=np.float32)
    ex = ((mx.view(np.uint32) >> np.uint32(23)) & np.uint32(0xff)).astype(np.int64)
    ex = np.maximum(ex, 15)
    scale = ((268 - ex).astype(np.uint32) << np.uint32(23)).view(np.float32)
    inv = ((ex - 14).astype(np.uint32) << np.uint32(23)).view(np.float32)
    return scale, inv


def split_scaled(x, scale):
    xs = (np.asarray(x, dtype=np.float32) * scale).astype(np.float32)
    with np.errstate(over="ignore"):
        h = xs.astype(np.float16)
        l = (xs - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h, l


def split(x):
    """x f32 [M, K] (K % 128 == 0) -> (h, l fp16 [M, K], inv f32 [K/128, M]) as frcnn_h2_split writes them."""
    x = np.asarray(x, dtype=np.float32)
    M, K = x.shape
    assert K % KB == 0
    xb = x.reshape(M, K // KB, KB)
    scale, inv = block_scale(np.abs(xb).max(axis=2))                   # [M, K/128]
    h, l = split_scaled(xb, scale[:, :, None])
    return h.reshape(M, K), l.reshape(M, K), np.ascontiguousarray(inv.T)


def split_grouped(x, group):
    """Producers that write several rows together may give them ONE scale per 128-k block, from the group's common maximum (the
    Winograd transforms: csrc/h2_common.h h2_emit_rows32 / 64).  x f32 [M, K], group int [M] (rows with equal id share) ->
    (h, l, inv) like split()."""
    x = np.asarray(x, dtype=np.float32)
    M, K = x.shape
    assert K % KB == 0
    xb = np.abs(x).reshape(M, K // KB, KB).max(axis=2)                 # [M, K/128]
    group = np.asarray(group)
    order = np.argsort(group, kind="stable")
    uniq, start = np.unique(group[order], return_index=True)
    gmax = np.maximum.reduceat(xb[order], start, axis=0)               # [groups, K/128]
    mx = np.empty_like(xb)
    mx[order] = np.repeat(gmax, np.diff(np.append(start, M)), axis=0)
    scale, inv = block_scale(mx)
    h, l = split_scaled(x.reshape(M, K // KB, KB), scale[:, :, None])
    return h.reshape(M, K), l.reshape(M, K), np.ascontiguousarray(inv.T)


def pack_w(w):
    """w f32 [G, N, K] -> (h, l fp16 [G, N, K], w_inv f32 [G, N]): one scale per output row (frcnn_h2_pack_w)."""
    w = np.asarray(w, dtype=np.float32)
    scale, inv = block_scale(np.maximum(np.abs(w).max(axis=2), np.float32(2.0 ** -40)))      # e_w <= 54: (bias + res) * 2^e_w stays finite
    h, l = split_scaled(w, scale[:, :, None])
    return h, l, inv


def gemm_terms(x, w, terms=3):
    """What the kernel's MFMAs evaluate, with exact (float64) accumulation: per 128-k block the kept cross terms of the scaled pieces,
    folded by the block's scale; finally times the weight row's scale.  x [M, K], w [N, K] -> float64 [M, N]."""
    xh, xl, xinv = split(x)
    wh, wl, winv = pack_w(w[None])
    wh, wl, winv = wh[0].astype(np.float64), wl[0].astype(np.float64), winv[0].astype(np.float64)
    xh, xl = xh.astype(np.float64), xl.astype(np.float64)
    M, K = x.shape
    out = np.zeros((M, w.shape[0]), dtype=np.float64)
    for kb in range(K // KB):
        s = slice(kb * KB, (kb + 1) * KB)
        blk = xh[:, s] @ wh[:, s].T + xh[:, s] @ wl[:, s].T + xl[:, s] @ wh[:, s].T
        if terms == 4:
            blk = blk + xl[:, s] @ wl[:, s].T
        out += blk * xinv[kb].astype(np.float64)[:, None]
    return out * winv[None, :]


def wgrad_terms(dy, x, slab=64):
    """What csrc/wgrad_h2.hip evaluates for dW = dY^T X (dY [M, Cout], X [M, K] float32, the reduction over the M pixels), with exact
    (float64) accumulation: every (column, `slab`-pixel segment) of both operands gets its own block_scale from the segment's
    largest magnitude (rows past M count as zeros), the scaled values are split into two fp16 pieces, a slab's product keeps the
    three leading cross terms, and the slab's sum is folded by the two inverse scales.  -> float64 [Cout, K]."""
    dy, x = np.asarray(dy, dtype=np.float32), np.asarray(x, dtype=np.float32)
    M = dy.shape[0]
    out = np.zeros((dy.shape[1], x.shape[1]), dtype=np.float64)
    for m0 in range(0, M, slab):
        a, b = dy[m0:m0 + slab], x[m0:m0 + slab]
        sa, ia = block_scale(np.abs(a).max(axis=0))                     # per column of dY
        sb, ib = block_scale(np.abs(b).max(axis=0))                     # per column of X
        ah, al = split_scaled(a, sa[None, :])
        bh, bl = split_scaled(b, sb[None, :])
        ah, al, bh, bl = (t.astype(np.float64) for t in (ah, al, bh, bl))
        blk = ah.T @ bh + al.T @ bh + ah.T @ bl
        out += blk * ia.astype(np.float64)[:, None] * ib.astype(np.float64)[None, :]
    return out
