"""CPU: the arithmetic behind csrc/gemm_x3.hip (cfg.HIP.MFMA_X3), restated in numpy -- the three-piece split is exact, every kept
cross term is an exact product of 8-bit significands, what the six-term product drops is bounded by 2^-24 |a w| (one f32 rounding of
the product) and unbiased; a truncating split (the first version) is 8x looser and biased -- plus the host rule that decides which
launches use it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd"), os.path.join(ROOT, "tf-faster-rcnn_amd", "lib")]


def bf16_rne(v):
    """float32 -> the nearest bf16 value (ties to even), as float32: what v_cvt_pk_bf16_f32 computes (finite inputs)."""
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_trunc(v):
    return (np.asarray(v, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x, to_bf16=bf16_rne):
    """x (float32) -> (h, m, l, s): the statement of x3_split8 / k_x3_pack; s = x - h - m before its conversion to bf16."""
    x = np.asarray(x, dtype=np.float32)
    h = to_bf16(x)
    r = (x - h).astype(np.float32)            # exact: at most 16 significant bits
    m = to_bf16(r)
    s = (r - m).astype(np.float32)            # exact: at most 8 significant bits are left
    return h, m, to_bf16(s), s


def _operands(n, seed):
    rng = np.random.RandomState(seed)
    return (rng.randn(n) * np.exp(rng.uniform(-10, 10, size=n))).astype(np.float32)


def test_three_bf16_pieces_are_an_exact_split():
    rng = np.random.RandomState(0)
    x = (rng.randn(200000) * np.exp(rng.uniform(-40, 40, size=200000))).astype(np.float32)
    for conv in (bf16_rne, bf16_trunc):
        h, m, l, s = split3(x, conv)
        assert np.array_equal(l, s)                                            # the third piece needs no rounding: nothing is lost
        assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
        for p in (h, m, l):                                                    # each piece is a bf16 value: low 16 bits clear
            assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    h, m, l, _ = split3(x)
    assert np.all(np.abs(m) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(l) <= np.abs(x) * 2.0 ** -17)


def test_six_term_product_drops_at_most_one_f32_rounding():
    a, w = _operands(1000000, 1), _operands(1000000, 2)
    exact = a.astype(np.float64) * w.astype(np.float64)
    stats = {}
    for name, conv in (("rne", bf16_rne), ("trunc", bf16_trunc)):
        ah, am, al, _ = [p.astype(np.float64) for p in split3(a, conv)]
        wh, wm, wl, _ = [p.astype(np.float64) for p in split3(w, conv)]
        six = ah * wh + (ah * wm + am * wh) + (ah * wl + al * wh + am * wm)    # float64 holds every 8 x 8-bit product and these sums exactly
        assert np.array_equal(six + (am * wl + al * wm + al * wl), exact)       # all nine cross terms: the product itself
        rel = (six - exact) / exact
        stats[name] = (np.abs(rel).max(), np.abs(rel).mean(), rel.mean())
        if name == "rne":                                                       # every kept term is exact in float32 (<= 16 significant bits)
            for t in (ah * wh, ah * wm, am * wh, ah * wl, al * wh, am * wm):
                assert np.array_equal(t.astype(np.float32).astype(np.float64), t)
    mx, mean, bias = stats["rne"]
    assert mx <= 2.0 ** -24 * 1.01 and mean <= 2.0 ** -27 and abs(bias) <= 2.0 ** -33        # <= one f32 rounding, unbiased
    f32_round = np.abs(exact.astype(np.float32).astype(np.float64) - exact) / np.abs(exact)
    assert mean <= f32_round.mean()                                             # on average below rounding the product to f32
    tmx, tmean, tbias = stats["trunc"]                                          # why the split rounds to nearest
    assert tmx > 4 * mx and tmean > 4 * mean and abs(tbias) > 0.9 * tmean       # truncation: looser, and every error towards zero


def test_host_rule_for_x3_launches():
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    net = resnetv1(101)
    net._mode = "TEST"
    old = cfg.HIP.MFMA_X3
    try:
        cfg.HIP.MFMA_X3 = True
        ok = net._x3_eligible
        assert ok(4 * 2394, 256, 1024, 1)              # block3 conv1, 4 images: 75 x 2 = 150 tiles
        assert not ok(2394, 256, 1024, 1)              # one image: 38 tiles -> the split-K f32 launch
        assert ok(2394, 1024, 256, 1)                  # block3 conv3, one image: 152 tiles
        assert ok(4 * 14700, 512, 2048, 1) and ok(1200, 512, 512, 121)          # block4 conv1, 7x7 Winograd products
        assert ok(150000, 64, 256, 1)                  # Cout = 64: 128 x 64 tiles
        assert not ok(9576, 18, 512, 1) and not ok(9576, 36, 512, 1)            # RPN heads: Cout not a multiple of 64
        assert not ok(9576, 256, 1000, 1)              # K % 32
        net._mode = "TRAIN"
        assert not ok(4 * 14700, 512, 2048, 1)         # filters change every step
        net._mode = "TEST"
        cfg.HIP.MFMA_X3 = False
        assert not ok(4 * 14700, 512, 2048, 1)
    finally:
        cfg.HIP.MFMA_X3 = old
