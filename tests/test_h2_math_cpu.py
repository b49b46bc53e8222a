"""CPU: the arithmetic behind csrc/gemm_h2.hip (cfg.HIP.MFMA_H2) stated in numpy (oracle/h2_ref.py): the block scale is an exact
power of two that puts the block maximum in [2^14, 2^15); h + l reproduces the scaled value to 2^-22 relative (or the fp16 subnormal
spacing for elements far below the block maximum); each kept cross term is an exact product of 11-bit significands; what the
three-term product drops is <= 3 * 2^-22 |x w|, unbiased, and at GEMM level sits at ~1e-7 of the output scale -- below the float32
accumulation noise of a plain f32 GEMM on the same data."""
import numpy as np

import h2_ref


def _acts(M, K, seed, kind):
    rng = np.random.RandomState(seed)
    if kind == "relu":
        a = np.maximum(rng.randn(M, K), 0) * np.exp(rng.uniform(-2, 2, size=(M, K)))
    elif kind == "wide":
        a = rng.randn(M, K) * np.exp(rng.uniform(-12, 12, size=(M, K)))
    else:
        a = rng.randn(M, K) * 3
    return a.astype(np.float32)


def test_block_scale_is_an_exact_power_of_two_and_centres_the_block():
    rng = np.random.RandomState(0)
    mx = np.abs(rng.randn(100000) * np.exp(rng.uniform(-60, 60, size=100000))).astype(np.float32)
    mx = mx[(mx > 2.0 ** -110) & (mx < 2.0 ** 120)]
    scale, inv = h2_ref.block_scale(mx)
    assert np.all(scale * inv == 1.0)
    assert np.all((scale.view(np.uint32) & np.uint32(0x7fffff)) == 0) and np.all((inv.view(np.uint32) & np.uint32(0x7fffff)) == 0)
    s = mx.astype(np.float64) * scale.astype(np.float64)
    assert np.all(s >= 2.0 ** 14) and np.all(s < 2.0 ** 15)
    z, zi = h2_ref.block_scale(np.zeros(3, dtype=np.float32))                  # all-zero block: clamped, finite
    assert np.all(np.isfinite(z)) and np.all(z * zi == 1.0)


def test_two_fp16_pieces_reproduce_the_value_to_2_pow_minus_22():
    for kind in ("relu", "wide", "randn"):
        x = _acts(64, 512, 1, kind)
        h, l, inv = h2_ref.split(x)
        assert h.dtype == np.float16 and l.dtype == np.float16 and np.all(np.isfinite(h.astype(np.float32)))
        rec = (h.astype(np.float64) + l.astype(np.float64)) * np.repeat(inv.T.astype(np.float64), 128, axis=1)
        err = np.abs(rec - x.astype(np.float64))
        blockmax = np.repeat(np.abs(x).reshape(64, 4, 128).max(axis=2), 128, axis=1).astype(np.float64)
        # 2^-22 of the value, or the fp16 subnormal spacing (2^-24 at the scaled magnitude = 2^-39 of the block maximum) if larger
        assert np.all(err <= np.maximum(np.abs(x) * 2.0 ** -22, blockmax * 2.0 ** -38))
        big = np.abs(x) >= blockmax * 2.0 ** -10
        assert np.all(err[big] <= np.abs(x[big]) * 2.0 ** -22)


def test_three_term_product_error_is_bounded_and_unbiased():
    rng = np.random.RandomState(2)
    a = (rng.randn(1, 128 * 2000) * np.exp(rng.uniform(-1, 1, size=(1, 128 * 2000)))).astype(np.float32)
    w = (rng.randn(1, 128 * 2000) * np.exp(rng.uniform(-1, 1, size=(1, 128 * 2000)))).astype(np.float32)
    ah, al, ainv = h2_ref.split(a)
    wh, wl, winv = h2_ref.split(w)
    sc = np.repeat(ainv.T.astype(np.float64), 128, axis=1) * np.repeat(winv.T.astype(np.float64), 128, axis=1)
    f = lambda t: t.astype(np.float64)
    three = (f(ah) * f(wh) + f(ah) * f(wl) + f(al) * f(wh)) * sc
    exact = f(a) * f(w)
    rel = (three - exact) / np.abs(exact)
    amax = np.repeat(np.abs(a).reshape(1, -1, 128).max(axis=2), 128, axis=1).astype(np.float64)
    wmax = np.repeat(np.abs(w).reshape(1, -1, 128).max(axis=2), 128, axis=1).astype(np.float64)
    big = (np.abs(a) >= amax * 2.0 ** -6) & (np.abs(w) >= wmax * 2.0 ** -6)    # both low pieces in fp16's normal range
    assert np.abs(rel[big]).max() <= 3.0 * 2.0 ** -22
    # elements far below their block's maximum: absolute floor 2^-38 of that maximum per operand (fp16 subnormal spacing, scaled back)
    assert np.all(np.abs(three - exact) <= 3.0 * 2.0 ** -22 * np.abs(exact) + 2.0 ** -37 * (amax * np.abs(w) + wmax * np.abs(a)))
    assert abs(rel[big].mean()) <= 2.0 ** -27                                     # round to nearest: no systematic sign
    assert np.sqrt((rel[big] ** 2).mean()) <= 2.0 ** -22.5
    # every kept term is exact in float32: 11-bit x 11-bit significands
    assert np.array_equal((ah.astype(np.float32) * wh.astype(np.float32)).astype(np.float64), f(ah) * f(wh))


def test_gemm_level_error_is_below_f32_accumulation_noise():
    rng = np.random.RandomState(3)
    for name, (M, K, N, kind) in {"block4 conv1": (96, 2048, 64, "relu"), "block4 conv3": (96, 512, 64, "relu"),
                                  "winograd": (96, 512, 64, "randn"), "wide": (96, 1024, 64, "wide")}.items():
        a = _acts(M, K, 4, kind)
        w = (rng.randn(N, K) / np.sqrt(K) * np.exp(rng.uniform(-1, 1, size=(N, K)))).astype(np.float32)
        exact = a.astype(np.float64) @ w.astype(np.float64).T
        scale = np.abs(exact).max()
        e3 = np.abs(h2_ref.gemm_terms(a, w, 3) - exact).max() / scale
        f32 = np.abs((a @ w.T).astype(np.float64) - exact).max() / scale           # numpy's float32 GEMM: one f32 implementation
        assert e3 <= 2.5e-7, (name, e3)
        assert e3 <= f32, (name, e3, f32)


def test_row_groups_sharing_a_scale_keep_the_representation_bounds():
    """The Winograd transforms give the rows one thread writes together ONE scale per 128-k block (their common maximum): every
    element is still h + l to 2^-22 of its value, or 2^-38 of the GROUP's block maximum where l leaves the fp16 normal range; a
    GEMM on such planes stays at the 1e-7 level."""
    rng = np.random.RandomState(7)
    M, K, N = 96, 512, 64
    x = (rng.randn(M, K) * np.exp(rng.uniform(-2, 2, size=(M, 1)))).astype(np.float32)          # rows of different magnitude
    group = np.arange(M) // 6                                                                  # six rows share (F(4x4,3x3) input rows)
    h, l, inv = h2_ref.split_grouped(x, group)
    assert np.all(np.isfinite(h.astype(np.float32)))
    for g in range(M // 6):
        assert np.all(inv[:, 6 * g:6 * g + 6] == inv[:, 6 * g:6 * g + 1])
    scaled_max = np.abs(x).reshape(M // 6, 6, K // 128, 128).max(axis=(1, 3)) / inv.T[::6].astype(np.float64)
    assert np.all(scaled_max >= 2.0 ** 14) and np.all(scaled_max < 2.0 ** 15)
    rec = (h.astype(np.float64) + l.astype(np.float64)) * np.repeat(inv.T.astype(np.float64), 128, axis=1)
    gmax = np.repeat(np.repeat(np.abs(x).reshape(M // 6, 6, K // 128, 128).max(axis=(1, 3)), 6, axis=0), 128, axis=1)
    assert np.all(np.abs(rec - x) <= np.maximum(np.abs(x) * 2.0 ** -22, gmax * 2.0 ** -38))
    w = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    wh, wl, winv = h2_ref.pack_w(w[None])
    f = lambda t: t.astype(np.float64)
    out = np.zeros((M, N))
    for kb in range(K // 128):
        s_ = slice(kb * 128, (kb + 1) * 128)
        out += (f(h[:, s_]) @ f(wh[0][:, s_]).T + f(h[:, s_]) @ f(wl[0][:, s_]).T + f(l[:, s_]) @ f(wh[0][:, s_]).T) * f(inv[kb])[:, None]
    out *= f(winv[0])[None, :]
    exact = f(x) @ f(w).T
    assert np.abs(out - exact).max() / np.abs(exact).max() <= 2.5e-7


def test_domain_outliers_inside_a_scale_block():
    """The one input pattern the format is NOT float32-class on, stated with numbers: a block scale follows the block's largest
    element, so elements more than ~2^20 below an outlier IN THE SAME 128-k block lose relative precision (absolute error 2^-38 of
    the outlier).  Up to a ratio of 1e6 the GEMM error stays below 1e-6 of the output scale; at 1e8 it is ~1e-4.  (The exact bf16x3
    split of csrc/gemm_x3.hip has no such limit; post-ReLU / batch-normalised activations span 2-4 decades per pixel.)"""
    rng = np.random.RandomState(0)
    M, K, N = 64, 512, 64
    errs = {}
    for ratio in (1e2, 1e4, 1e6, 1e8):
        a = np.maximum(rng.randn(M, K), 0).astype(np.float32)
        a[:, ::128] = ratio                                   # one outlier per block ...
        w = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
        w[:, ::128] = 0                                       # ... that the result does not depend on
        exact = a.astype(np.float64) @ w.astype(np.float64).T
        errs[ratio] = np.abs(h2_ref.gemm_terms(a, w, 3) - exact).max() / np.abs(exact).max()
    assert errs[1e2] <= 3e-7 and errs[1e4] <= 3e-7 and errs[1e6] <= 2e-6
    assert 1e-6 < errs[1e8] < 1e-3                            # documented limit, not an accident


def test_weight_gradient_in_two_piece_fp16_is_below_float32_accumulation_noise():
    """csrc/wgrad_h2.hip in numpy (h2_ref.wgrad_terms): dW = dY^T X over 2394 pixels (a 38 x 63 map) with post-ReLU activations and
    gated gradients whose magnitude varies over the map; scales per (channel, 64-pixel segment), three cross terms.  With exact
    accumulation the format's own error is ~1e-7 of max |dW| -- below what a float32 dot product of this length accumulates."""
    rng = np.random.RandomState(11)
    M, Cout, Cin = 2394, 48, 64
    x = (np.maximum(rng.randn(M, Cin), 0) * np.exp(rng.uniform(-2, 2, size=(M, 1)))).astype(np.float32)
    dy = (rng.randn(M, Cout) * (rng.rand(M, Cout) < 0.4) * np.exp(rng.uniform(-3, 3, size=(M, 1)))).astype(np.float32)
    exact = dy.astype(np.float64).T @ x.astype(np.float64)
    got = h2_ref.wgrad_terms(dy, x)
    err = np.abs(got - exact).max() / np.abs(exact).max()
    f32 = np.zeros((Cout, Cin), dtype=np.float32)                                          # a float32 accumulation in pixel order
    for m0 in range(0, M, 2):                                                              # (k = 2 per f32 MFMA)
        f32 = (f32 + dy[m0:m0 + 2].T @ x[m0:m0 + 2]).astype(np.float32)
    err32 = np.abs(f32.astype(np.float64) - exact).max() / np.abs(exact).max()
    print("wgrad h2 format error %.2e, float32 accumulation %.2e" % (err, err32))
    assert err <= 2e-7 and err <= err32
    # a whole segment of zeros in ONE operand (rows past M, a padding tap) contributes exactly nothing, whatever the other holds
    d0, x0 = dy[:37 * 64], x[:37 * 64]
    base = h2_ref.wgrad_terms(d0, x0)
    more = h2_ref.wgrad_terms(np.vstack([d0, np.zeros((64, Cout), np.float32)]), np.vstack([x0, 7 * np.ones((64, Cin), np.float32)]))
    assert np.array_equal(base, more)
