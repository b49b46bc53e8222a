"""GPU parity of csrc/gemm_h2.hip through the C ABI: the splitters reproduce the numpy statement of the operand format
(oracle/h2_ref.py) bit for bit; frcnn_gemm_h2 against float64 is (a) inside the dense-kernel bound 2e-5 of the output scale and (b)
of the f32-MFMA kernel's class on the same data (<= its error on the path's shapes, <= 3x in general); the planes emitted from the
GEMM epilogue are bit-identical to frcnn_h2_split of the f32 result; rows past M untouched."""
import numpy as np
import pytest
import torch

import h2_ref

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _planes_np(h2):
    raw = h2.planes.cpu().numpy().view(np.float16).reshape(2, h2.rows, h2.K)
    return raw[0], raw[1], h2.inv.cpu().numpy()


@pytest.mark.parametrize("M,K,kind", [(333, 256, "randn"), (1200, 512, "relu"), (77, 1024, "wide"), (64, 128, "zeros")])
def test_h2_split_matches_numpy_statement(dev, M, K, kind):
    from frcnn_hip import ops
    rng = np.random.RandomState(M + K)
    if kind == "relu":
        x = np.maximum(rng.randn(M, K), 0) * np.exp(rng.uniform(-3, 3, size=(M, K)))
    elif kind == "wide":
        x = rng.randn(M, K) * np.exp(rng.uniform(-30, 30, size=(M, K)))
    elif kind == "zeros":
        x = np.zeros((M, K))
        x[3, 5], x[7, :] = 1e-30, 65504.0 * 3
    else:
        x = rng.randn(M, K) * 3
    x = x.astype(np.float32)
    got = _planes_np(ops.h2_split(T(x, dev)))
    want = h2_ref.split(x)
    for g, w_, name in zip(got, want, ("h", "l", "inv")):
        assert np.array_equal(g.view(np.uint16) if g.dtype == np.float16 else g, w_.view(np.uint16) if w_.dtype == np.float16 else w_), name


def test_h2_pack_w_matches_numpy_statement(dev):
    from frcnn_hip import ops
    rng = np.random.RandomState(5)
    G, N, K = 3, 128, 320
    w = (rng.randn(G, N, K) * np.exp(rng.uniform(-8, 8, size=(G, N, 1)))).astype(np.float32)
    w[0, 3, :] = 0.0                                                            # an all-zero filter row: the scale stops at 2^54
    w[1, 7, :] *= np.float32(1e-20)
    planes, winv = ops.h2_pack_w(T(w, dev))
    raw = planes.cpu().numpy().view(np.float16).reshape(G, 2, N, K)
    h, l, inv = h2_ref.pack_w(w)
    assert np.array_equal(raw[:, 0].view(np.uint16), h.view(np.uint16)) and np.array_equal(raw[:, 1].view(np.uint16), l.view(np.uint16))
    assert np.array_equal(winv.cpu().numpy(), inv)


H2_CASES = [
    # G, M, N, K, residual, act
    (1, 128 * 40 + 76, 256, 128, True, 1),       # conv3-like, M tail
    (1, 128 * 9 + 4, 128, 2048, False, 1),       # long K: 16 scale blocks
    (5, 64 * 9 + 12, 128, 128, False, 0),        # batched product (Winograd), M tail inside every batch entry
    (1, 128 * 70, 2048, 128, True, 2),           # 1120 tiles, ReLU6
    (1, 128 * 33 + 8, 128, 256, True, 1),
    (3, 1200, 512, 512, False, 0),               # the 7x7 Winograd product shape (fewer points)
    (1, 2394, 256, 1024, False, 1),              # one 38 x 63 image: M % 4 == 2
    (7, 131, 128, 256, False, 0),                # batched, M odd: every batch entry starts at an unaligned scale row
    (1, 61, 128, 128, True, 1),                  # less than one tile
]


@pytest.mark.parametrize("cfg", [9, 12, 21, -1])
@pytest.mark.parametrize("case", H2_CASES, ids=[str(i) for i in range(len(H2_CASES))])
def test_gemm_h2_is_f32_class(dev, case, cfg):
    from frcnn_hip import ops
    G, M, N, K, with_res, act = case
    rng = np.random.RandomState(M % 997 + N + K)
    x = rng.randn(G * M, K).astype(np.float32)
    x[:, ::7] *= 1e3                                                            # mixed magnitudes inside every scale block
    x[::5] *= 1e-4                                                              # and across rows
    w = (rng.randn(G, N, K) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32) if G == 1 else None
    r = rng.randn(G * M, N).astype(np.float32) if with_res else None
    want = np.einsum("gmk,gnk->gmn", x.reshape(G, M, K).astype(np.float64), w.astype(np.float64)).reshape(G * M, N)
    if b is not None: want += b.astype(np.float64)
    if r is not None: want += r.astype(np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    if act == 1: want = np.maximum(want, 0)
    if act == 2: want = np.clip(want, 0, 6)
    xd, wd = T(x, dev), T(w, dev)
    bd, rd = (T(b, dev) if b is not None else None), (T(r, dev) if r is not None else None)
    xp, wp = ops.h2_split(xd), ops.h2_pack_w(wd)
    guard = torch.full((G * M + 64, N), 7.25, dtype=torch.float32, device=dev)
    yp = ops.H2.empty(G * M, N, dev)
    ops.gemm_h2(xp, wp, G, M, N, K, bd, rd, act, out=guard[:G * M], out_planes=yp, cfg=cfg)
    torch.cuda.synchronize()
    got = guard[:G * M].cpu().numpy()
    assert bool((guard[G * M:] == 7.25).all()), "store past the last row"
    if G == 1:
        f32 = ops.conv2d(xd.view(1, 1, M, K), wd.view(N, 1, 1, K), bd, 1, 1, 1, (0, 0, 0, 0), act, None if rd is None else rd.view(1, 1, M, N), 1).view(M, N).cpu().numpy()
    else:
        f32 = ops.gemm_batched_nt(xd.view(G, M, K), wd, torch.empty((G, M, N), dtype=torch.float32, device=dev)).view(G * M, N).cpu().numpy()
    e_h2, e_f32 = float(np.abs(got - want).max()) / scale, float(np.abs(f32 - want).max()) / scale
    print("h2 cfg %d %s: max |err| / scale = %.3e (f32-MFMA kernel %.3e)" % (cfg, str(case), e_h2, e_f32))
    assert e_h2 <= 2e-5 and e_h2 <= 3.0 * e_f32 + 1e-7
    # the planes emitted by the epilogue == the splitter applied to the f32 result, bit for bit
    want_p = _planes_np(ops.h2_split(guard[:G * M].contiguous()))
    got_p = _planes_np(yp)
    for g_, w_, name in zip(got_p, want_p, ("h", "l", "inv")):
        assert np.array_equal(g_.view(np.uint16) if g_.dtype == np.float16 else g_, w_.view(np.uint16) if w_.dtype == np.float16 else w_), name


def test_gemm_h2_planes_only_and_chained(dev):
    """y = None: only the operand planes are written; feeding them to the next GEMM equals feeding the split of the f32 result."""
    from frcnn_hip import ops
    rng = np.random.RandomState(1)
    M, K, N, N2 = 128 * 5 + 20, 256, 256, 128
    x = np.maximum(rng.randn(M, K), 0).astype(np.float32)
    w1 = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    w2 = (rng.randn(N2, N) / np.sqrt(N)).astype(np.float32)
    xp, w1p, w2p = ops.h2_split(T(x, dev)), ops.h2_pack_w(T(w1, dev)), ops.h2_pack_w(T(w2, dev))
    y1, _ = ops.gemm_h2(xp, w1p, 1, M, N, K, act=1)
    yp = ops.H2.empty(M, N, dev)
    none, _ = ops.gemm_h2(xp, w1p, 1, M, N, K, act=1, out_planes=yp, want_f32=False)
    assert none is None
    a, _ = ops.gemm_h2(yp, w2p, 1, M, N2, N)
    b, _ = ops.gemm_h2(ops.h2_split(y1), w2p, 1, M, N2, N)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    want = np.maximum(x.astype(np.float64) @ w1.astype(np.float64).T, 0) @ w2.astype(np.float64).T
    assert float(np.abs(a.cpu().numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))


def test_gemm_h2_residual_as_planes(dev):
    """The residual given as operand planes (cfg.HIP.H2_TRUNK_PLANES) is read as (h + l) * 2^-e: identical to passing that float32
    tensor, for tiles with an M tail, batched entries and both tile shapes."""
    from frcnn_hip import ops
    rng = np.random.RandomState(8)
    for G, M, N, K, cfg_ in ((1, 128 * 5 + 37, 256, 128, 9), (1, 9576, 1024, 256, 12), (3, 211, 128, 256, 9), (1, 2394, 1024, 256, -1), (1, 256 * 9 + 50, 256, 1024, 21)):
        x = np.maximum(rng.randn(G * M, K), 0).astype(np.float32)
        w = (rng.randn(G, N, K) / np.sqrt(K)).astype(np.float32)
        r = (np.maximum(rng.randn(G * M, N), 0) * np.exp(rng.uniform(-2, 2, size=(G * M, 1)))).astype(np.float32)
        b = T(rng.randn(N).astype(np.float32), dev) if G == 1 else None
        xp, wp, rp = ops.h2_split(T(x, dev)), ops.h2_pack_w(T(w, dev)), ops.h2_split(T(r, dev))
        y1, _ = ops.gemm_h2(xp, wp, G, M, N, K, b, rp, 1, cfg=cfg_)
        y2, _ = ops.gemm_h2(xp, wp, G, M, N, K, b, rp.to_float().contiguous(), 1, cfg=cfg_)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2), (G, M, N, K)
        want = np.einsum("gmk,gnk->gmn", x.reshape(G, M, K).astype(np.float64), w.astype(np.float64)).reshape(G * M, N) + r
        if b is not None: want = want + b.cpu().numpy().astype(np.float64)
        want = np.maximum(want, 0)
        assert float(np.abs(y1.cpu().numpy() - want).max()) <= 1e-6 * max(1.0, float(np.abs(want).max()))


def _same_planes(got, want):
    """got: ops.H2 on the device; want: an ops.H2 or the (h, l, inv) arrays of the numpy statement"""
    want = _planes_np(want) if hasattr(want, "planes") else want
    for g_, w_, name in zip(_planes_np(got), want, ("h", "l", "inv")):
        assert np.array_equal(g_.view(np.uint16) if g_.dtype == np.float16 else g_, w_.view(np.uint16) if w_.dtype == np.float16 else w_), name


def _wino_groups(m, N, H, W):
    """row -> scale-group id of the plane-emitting transforms: the rows ONE thread writes together share a scale (csrc/winograd*.hip):
    input side = the (m+2) / 11 points of one transform row of one tile; output side = the pixels of one output row of one tile."""
    from frcnn_hip import ops
    G, Tl = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
    a = 11 if m == 7 else m + 2
    xn, t = np.divmod(np.arange(G * Tl), Tl)
    gin = (xn // a) * Tl + t
    if m == 7:
        gout = np.arange(N * 49) // 7
    else:
        TW = (W + m - 1) // m
        img, rem = np.divmod(np.arange(N * H * W), H * W)
        oh, ow = np.divmod(rem, W)
        gout = (img * H + oh) * TW + ow // m
    return gin, gout


@pytest.mark.parametrize("m,N,H,W,C", [(2, 2, 19, 25, 128), (4, 3, 38, 63, 256), (4, 1, 10, 13, 384), (7, 37, 7, 7, 512), (7, 4, 7, 7, 128)])
def test_winograd_transforms_emit_the_planes_of_their_f32_results(dev, m, N, H, W, C):
    """frcnn_winograd[7]_input/output_transform_h2: V / y as operand planes == the numpy statement of the format applied to the float32
    transform (row-group scales, bit for bit); the optional float32 output of the output transform == the plain transform's."""
    from frcnn_hip import ops
    rng = np.random.RandomState(m * 100 + C)
    x = (np.maximum(rng.randn(N, H, W, C), 0) * np.exp(rng.uniform(-3, 3, size=(N, H, W, C)))).astype(np.float32)
    x[:, :, :, 5] = 0.0                                                  # a dead channel
    x[0, 0:6, 0:6, :] = 0.0                                              # all-zero tiles: whole scale groups of zeros
    G, Tl = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
    gin, gout = _wino_groups(m, N, H, W)
    xd = T(x, dev)
    v = torch.empty((G, Tl, C), dtype=torch.float32, device=dev)
    ops.winograd_input_transform(xd, v, m)
    vp = ops.H2.empty(G * Tl, C, dev)
    vp.planes.fill_(0x5a); vp.inv.fill_(-1.0)
    ops.winograd_input_transform_h2(xd, vp, m)
    torch.cuda.synchronize()
    _same_planes(vp, h2_ref.split_grouped(v.view(G * Tl, C).cpu().numpy(), gin))
    mm = torch.from_numpy((rng.randn(G, Tl, C) * 3).astype(np.float32)).to(dev)
    bias = T(rng.randn(C).astype(np.float32), dev)
    for act in (0, 1):
        y = torch.empty((N, H, W, C), dtype=torch.float32, device=dev)
        ops.winograd_output_transform(mm, bias, act, y, m)
        yp, y2 = ops.H2.empty(N * H * W, C, dev), torch.full((N, H, W, C), 7.0, dtype=torch.float32, device=dev)
        yp.planes.fill_(0x5a); yp.inv.fill_(-1.0)
        ops.winograd_output_transform_h2(mm, bias, act, (N, H, W, C), m, yp, y2)
        assert torch.equal(y, y2)
        _same_planes(yp, h2_ref.split_grouped(y.view(N * H * W, C).cpu().numpy(), gout))
        yp2 = ops.H2.empty(N * H * W, C, dev)
        ops.winograd_output_transform_h2(mm, bias, act, (N, H, W, C), m, yp2, None)          # planes only
        _same_planes(yp2, yp)
        # and the planes are a faithful image of the float32 tensor: 2^-22 of the value or 2^-38 of the group's block maximum
        rec = yp.to_float().double().cpu().numpy()
        want = y.view(N * H * W, C).double().cpu().numpy()
        assert np.all(np.abs(rec - want) <= np.maximum(np.abs(want) * 2.0 ** -22, np.abs(want).max() * 2.0 ** -30))


@pytest.mark.parametrize("N,H,W,C,stride", [(2, 19, 25, 128, 1), (1, 38, 63, 256, 2), (3, 7, 7, 512, 1)])
def test_depthwise_conv_emits_the_planes_of_its_f32_result(dev, N, H, W, C, stride):
    """frcnn_dwconv3x3_nhwc_h2: the depthwise result as operand planes == frcnn_h2_split of the float32 result (bit for bit), with and
    without the float32 tensor."""
    from frcnn_hip import ops
    rng = np.random.RandomState(C + stride)
    x = (rng.randn(N, H, W, C) * np.exp(rng.uniform(-2, 2, size=(1, 1, 1, C)))).astype(np.float32)
    w = rng.randn(3, 3, C).astype(np.float32)
    b = rng.randn(C).astype(np.float32)
    xd, wd, bd = T(x, dev), T(w, dev), T(b, dev)
    y = ops.dwconv3x3(xd, wd, bd, stride, (1, 1, 1, 1), 2)
    rows = y.numel() // C
    yp = ops.H2.empty(rows, C, dev)
    yp.planes.fill_(0x5a); yp.inv.fill_(-1.0)
    y2 = ops.dwconv3x3(xd, wd, bd, stride, (1, 1, 1, 1), 2, out_planes=yp, want_f32=True)
    assert torch.equal(y, y2)
    _same_planes(yp, ops.h2_split(y.view(rows, C)))
    yp2 = ops.H2.empty(rows, C, dev)
    assert ops.dwconv3x3(xd, wd, bd, stride, (1, 1, 1, 1), 2, out_planes=yp2, want_f32=False) is None
    _same_planes(yp2, yp)


@pytest.mark.parametrize("shape", [(1, 256 * 300 + 40, 512, 512, True), (1, 58800, 512, 2048, False), (121, 1200, 512, 512, False),
                                   (1, 9576, 1024, 256, True), (3, 300, 128, 128, False), (1, 9576, 1024, 256, "planes"),
                                   (1, 128 * 70 + 9, 2048, 512, "planes")],
                         ids=["many_tiles", "b4c1", "w7", "b3c3", "tiny", "b3c3_trunk_planes", "b4c3_trunk_planes"])
@pytest.mark.parametrize("pp", [21, 12, 30, 31, 32, 33, 34, 40, 41, -1, -4, -8])
def test_gemm_h2_ping_pong_is_bit_identical_to_the_one_barrier_schedule(dev, shape, pp):
    """cfg 21 (256 x 128 tiles, two wave groups a segment apart, 3-slot ring), cfg 12 (64-row tiles), round 5's cfgs 30-33 (the light tile
    boundary: filter scales + bias through LDS, the residual raw in the accumulators until the first fold, counted vmcnt, 16-byte plane
    stores) and the by-shape choice (-1) multiply and fold in the same order as cfg 9: the f32 result, the emitted planes and the block
    scales must be the same BITS, on every one of several launches (a schedule with a race differs from launch to launch), with several
    tiles per resident workgroup and M tails -- incl. the conv3 class (K = 256 / 512, residual + f32 + planes) with the residual given as
    float32 or as operand planes (cfg.HIP.H2_TRUNK_PLANES, the default since round 5).  Round 6: cfgs 40 / 41 = the deferred epilogue
    (tile t's results leave under tile t + 1's first three slabs, the residual arrives raw in the accumulator's own registers), -8 = DE
    wherever it exists."""
    from frcnn_hip import ops
    G, M, N, K, with_res = shape
    torch.manual_seed(G + M)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 6 - 3)
    w = torch.randn(G, N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) if G == 1 else None
    r = torch.randn(G * M, N, device=dev) if with_res else None
    if with_res == "planes":
        r = ops.h2_split(r.clamp(min=0) * torch.exp(torch.rand(G * M, N, device=dev) * 4 - 2))
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    ref, refp = torch.empty(G * M, N, device=dev), ops.H2.empty(G * M, N, dev)
    ops.gemm_h2(xp, wp, G, M, N, K, b, r, 1, out=ref, out_planes=refp, cfg=9)
    torch.cuda.synchronize()
    for rep in range(6):
        got, gotp = torch.full((G * M, N), float("nan"), device=dev), ops.H2.empty(G * M, N, dev)
        gotp.planes.zero_(); gotp.inv.zero_()
        ops.gemm_h2(xp, wp, G, M, N, K, b, r, 1, out=got, out_planes=gotp, cfg=pp)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), "f32 result, launch %d" % rep
        assert torch.equal(gotp.planes.view(torch.int16), refp.planes.view(torch.int16)) and torch.equal(gotp.inv, refp.inv), "planes, launch %d" % rep


@pytest.mark.parametrize("form", ["planes_res_planes_out", "planes_res_f32_out", "f32_res_planes_out", "no_res_planes_out", "no_res_no_bias_f32"])
@pytest.mark.parametrize("shape", [(1, 128 * 70 + 9, 2048, 512), (1, 19152, 1024, 256), (1, 128 * 90 + 1, 512, 128), (4, 128 * 9 + 77, 256, 128),
                                   (1, 100, 128, 128), (2, 128, 128, 384)],
                         ids=["b4c3", "b3c3", "b2c3_four_slabs_per_tile", "batched_tails", "one_partial_tile", "one_tile_per_entry"])
def test_gemm_h2_deferred_epilogue_every_form_is_bit_identical(dev, shape, form):
    """The deferred epilogue (cfgs 40 / 41, csrc/gemm_h2.hip "TUNE & 1024") in the forms the network launches -- the SHIPPED one first:
    residual read as operand planes, planes only out (the identity units of a trunk kept as planes) -- against cfg 9 on the same
    operands: same f32 bits, same plane bits, same block scales, six launches each.  Covers K = 128 (a tile is exactly the four slabs
    the drain needs), one tile per workgroup (nothing to defer: the standalone epilogue alone), M tails inside batch entries (rows past
    M are dropped by the descriptor's range, loads return 0), and every residual kind through the one branch-free load stream."""
    from frcnn_hip import ops
    G, M, N, K = shape
    torch.manual_seed(M + K)
    x = torch.randn(G * M, K, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, K, device=dev) * 6 - 3)
    w = torch.randn(G, N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) if (G == 1 and "no_bias" not in form) else None
    r = None
    if form.startswith("planes_res"):
        r = ops.h2_split(torch.randn(G * M, N, device=dev).clamp(min=0) * torch.exp(torch.rand(G * M, N, device=dev) * 4 - 2))
    elif form.startswith("f32_res"):
        r = torch.randn(G * M, N, device=dev)
    f32_out, planes_out = form.endswith("f32_out") or form.endswith("f32"), form.endswith("planes_out")
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)

    def run(c):
        y = torch.full((G * M, N), float("nan"), device=dev) if f32_out else None
        yp = ops.H2.empty(G * M, N, dev) if planes_out else None
        if yp is not None:
            yp.planes.zero_(); yp.inv.zero_()
        ops.gemm_h2(xp, wp, G, M, N, K, b, r, 1, out=y, out_planes=yp, want_f32=f32_out, cfg=c)
        torch.cuda.synchronize()
        return y, yp
    ref, refp = run(9)
    for c in (40, 41, -1):
        for rep in range(6 if c > 0 else 2):
            y, yp = run(c)
            if f32_out:
                assert torch.equal(y.view(torch.int32), ref.view(torch.int32)), (c, rep)
            if planes_out:
                assert torch.equal(yp.planes.view(torch.int16), refp.planes.view(torch.int16)) and torch.equal(yp.inv, refp.inv), (c, rep)


@pytest.mark.parametrize("R,K,N,with_res", [(300, 512, 2048, True), (37, 128, 256, False), (48, 256, 128, True)])
def test_gemm_h2_mean_is_slot_invariant_and_matches_conv_then_mean(dev, R, K, N, with_res):
    """frcnn_gemm_h2_mean (the tail's last conv3 + residual + ReLU + reduce_mean over the 49 positions of a RoI, no [R*49, N] tensor):
    equals the float64 mean of frcnn_gemm_h2's float32 result to f32 summation noise; ONE batch entry per image makes the reduction
    order a function of the RoI's index inside its image -- the same image gives the same bits alone (G = 1), in slot 0 and in slot 2
    of a batch of three, and under every tile configuration."""
    from frcnn_hip import ops
    rng = np.random.RandomState(R + K)
    M = R * 49
    imgs = [np.maximum(rng.randn(M, K), 0).astype(np.float32) for _ in range(2)]
    ress = [rng.randn(M, N).astype(np.float32) for _ in range(2)]
    w = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32)
    wp, bd = ops.h2_pack_w(T(w[None], dev)), T(b, dev)

    def run(order, cfg):
        x = T(np.concatenate([imgs[i] for i in order], axis=0), dev)
        r = T(np.concatenate([ress[i] for i in order], axis=0), dev) if with_res else None
        return ops.gemm_h2_mean(ops.h2_split(x), wp, len(order), M, N, K, bd, r, 1, 49, cfg=cfg).cpu().numpy()
    one = run([0], 9)
    y, _ = ops.gemm_h2(ops.h2_split(T(imgs[0], dev)), wp, 1, M, N, K, bd, T(ress[0], dev) if with_res else None, 1)
    want = y.cpu().numpy().astype(np.float64).reshape(R, 49, N).mean(axis=1)
    err = float(np.abs(one - want).max()) / max(1.0, float(np.abs(want).max()))
    print("gemm_h2_mean R %d K %d N %d: max |err| / scale vs float64 mean of the f32 result = %.2e" % (R, K, N, err))
    assert one.shape == (R, N) and err <= 2e-6
    three = run([0, 1, 0], 9)
    assert np.array_equal(three[:R], one) and np.array_equal(three[2 * R:], one) and not np.array_equal(three[R:2 * R], one)
    for cfg in (12, 21, -1, 40, 41):              # (40 / 41: the fused-mean form keeps the standalone epilogue -- the ids map to 31 / 33)
        assert np.array_equal(run([0], cfg), one), cfg
