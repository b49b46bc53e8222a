"""The training step's data-gradient chain with its elementwise passes folded into the launches around them (frcnn_hip/train.py _sweep:
ReLU-gradient masks in the producing launch's epilogue, identity-shortcut gradients borrowed instead of copied, operand planes of dY out
of the Winograd output transform; lib/nets/resnet_v1.py's bottleneck in reverse).  Every mask is an exact select, so the float32 results
are BIT equal to the unfused sequence -- per C-ABI entry and over a whole training step.  Operand planes: the GEMM epilogue emits exactly
what frcnn_h2_split would; the Winograd transforms share one block scale between the rows a thread writes together (h2_common.h
h2_emit_rows32 / 64: any power of two that keeps the block below 2^15 is a valid scale), so their planes are checked as a representation
of the same tensor (<= 2^-21 of the block maximum), not as the splitter's bits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes_hold(yp, want):
    """yp (ops.H2) represents the float32 tensor `want` [rows, K] to the format's precision: |(h + l) 2^-e - x| <= 2^-21 max|block|."""
    rows, K = yp.rows, yp.K
    x = want.reshape(rows, K)
    err = (yp.to_float() - x).abs().reshape(rows, K // 128, 128).amax(dim=2)
    top = x.abs().reshape(rows, K // 128, 128).amax(dim=2)
    assert bool((err <= top * 2.0 ** -21 + 1e-37).all()), float((err / (top + 1e-37)).max())
    e = torch.log2(yp.inv)
    assert bool((e == e.round()).all())                                    # exact powers of two


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, k, residual            (M = N*H*W)
    (1, 38, 63, 1024, 256, 1, False),            # block3 conv3's data gradient: 38 tiles -> split-K, the finishing pass masks
    (1, 38, 63, 512, 256, 1, True),              # ... accumulating into a buffer that already holds a gradient
    (2, 75, 125, 128, 128, 3, False),            # 3x3, enough tiles for one pass: the kernel's own epilogue masks
    (1, 19, 32, 64, 24, 1, True),                # Cout % 4 == 0 but < 32 lanes; tiny
    (1, 7, 10, 32, 18, 1, False),                # scalar epilogue (Cout % 4 != 0)
])
def test_conv2d_masked_equals_conv2d_then_relu_bwd(dev, case):
    from frcnn_hip import ops
    N, H, W, Cin, Cout, k, with_res = case
    x = _rand((N, H, W, Cin), dev, 1)
    w = _rand((Cout, k, k, Cin), dev, 2, 0.05)
    fwd = _rand((N, H, W, Cout), dev, 3)                                   # the forward activation: about half of it <= 0
    fwd[0, 0, 0, 0] = 0.0                                                   # y == 0 is masked (y > 0 ? g : 0)
    res = _rand((N, H, W, Cout), dev, 4) if with_res else None
    pad = (k // 2,) * 4
    want = ops.conv2d(x, w, None, k, k, 1, pad, 0, res, 1)
    ops.relu_bwd(want, fwd)
    got = ops.conv2d(x, w, None, k, k, 1, pad, 0, res, 1, mask=fwd)
    assert torch.equal(got, want)
    assert float((got == 0).float().mean()) > 0.3
    if with_res:                                                            # in place: residual == out, as the sweep accumulates
        buf = res.clone()
        ops.conv2d(x, w, None, k, k, 1, pad, 0, buf, 1, out=buf, mask=fwd)
        assert torch.equal(buf, want)


@pytest.mark.parametrize("cfg", [-1, 9, 12, 21, 30, 31, 32, 33, 34, 40, 41])       # round 6: the light-boundary / deferred-epilogue configurations have masked twins
@pytest.mark.parametrize("shape", [(2394, 1024, 256), (12544, 2048, 512), (300, 128, 128)])
def test_gemm_h2_masked_equals_gemm_h2_then_relu_bwd(dev, shape, cfg):
    from frcnn_hip import ops
    M, N, K = shape
    x = _rand((M, K), dev, 5)
    w = _rand((N, K), dev, 6, 0.05)
    fwd = _rand((M, N), dev, 7)
    res = _rand((M, N), dev, 8)
    xp, wp = ops.h2_split(x), ops.h2_pack_w(w)
    for r in (None, res):
        want, _ = ops.gemm_h2(xp, wp, 1, M, N, K, None, r, 0, cfg=cfg)
        ops.relu_bwd(want, fwd)
        yp = ops.H2.empty(M, N, dev)
        got, _ = ops.gemm_h2(xp, wp, 1, M, N, K, None, r, 0, cfg=cfg, mask=fwd, out_planes=yp)
        assert torch.equal(got, want)
        ref = ops.h2_split(want)                                            # the emitted planes are those of the MASKED tensor
        assert torch.equal(yp.planes, ref.planes) and torch.equal(yp.inv, ref.inv)
    buf = res.clone()                                                       # in place, as the sweep accumulates into an existing gradient
    ops.gemm_h2(xp, wp, 1, M, N, K, None, buf, 0, out=buf, cfg=cfg, mask=fwd)
    assert torch.equal(buf, want)
    nanx = x.clone()
    nanx[3, 5] = float("nan")                                               # NaN passes where the mask is positive, like frcnn_relu_bwd
    fwd2 = fwd.clone()
    fwd2[3, :] = 1.0
    got, _ = ops.gemm_h2(ops.h2_split(nanx), wp, 1, M, N, K, None, None, 0, cfg=cfg, mask=fwd2)
    assert bool(torch.isnan(got[3]).all()) and not bool(torch.isnan(got[4:]).any())


@pytest.mark.parametrize("case", [(4, 1, 38, 63, 256), (4, 2, 19, 30, 128), (4, 1, 7, 5, 64), (7, 12, 7, 7, 512), (7, 3, 7, 7, 64)])
def test_winograd_output_masked_equals_transform_then_relu_bwd(dev, case):
    from frcnn_hip import ops
    m, N, H, W, C = case
    G, T = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
    mm = _rand((G, T, C), dev, 9)
    fwd = _rand((N, H, W, C), dev, 10)
    want = ops.winograd_output_transform(mm, None, 0, torch.empty((N, H, W, C), device=dev), m)
    ops.relu_bwd(want, fwd)
    got = ops.winograd_output_transform_masked(mm, fwd, torch.empty((N, H, W, C), device=dev), m)
    assert torch.equal(got, want)
    if C % 128 == 0:
        yp = ops.H2.empty(N * H * W, C, dev)
        got = ops.winograd_output_transform_masked(mm, fwd, torch.empty((N, H, W, C), device=dev), m, yp)
        assert torch.equal(got, want)
        _planes_hold(yp, want)


def test_training_forward_winograd_emits_float32_and_planes(dev):
    """TRAIN forward: conv2's output transform writes float32 + the operand planes of conv3 (lib/nets/network.py _conv, emit_h2)."""
    from frcnn_hip import ops
    x = _rand((1, 38, 63, 256), dev, 11)
    u = _rand((36, 256, 256), dev, 12, 0.05)
    b = _rand((256,), dev, 13)
    want = ops.conv3x3_winograd(x, u, b, 1)
    yp = ops.H2.empty(38 * 63, 256, dev)
    got = ops.conv3x3_winograd(x, u, b, 1, out_planes=yp)
    assert torch.equal(got, want)
    _planes_hold(yp, want)


def _one_step(dev, fuse, tag, min_tiles, steps=2, pipe=True):
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from frcnn_hip import ops
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    rng = np.random.RandomState(4)
    image = ((rng.rand(1, 224, 288, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [100, 70, 200, 150, 12]], dtype=np.float32)
    blobs = dict(data=image, im_info=np.array([224, 288, 1.0], dtype=np.float32), gt_boxes=gt)
    cfg.HIP.H2_MIN_TILES = min_tiles
    sess = Session(device=dev, seed=9)
    net = resnetv1(num_layers=50)
    net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
    sess.init_variables(net.variable_specs())
    ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4)
    ts.lr = 1e-3
    ts.fuse_chain = fuse
    ts.pipe_dgrads = pipe
    calls = {}
    real = ops.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return real(name, *a)
    ops.call = counting
    try:
        losses = [net.train_step(sess, blobs, ts)]
    finally:
        ops.call = real
    torch.cuda.synchronize()
    grads = {sc: p.grad_w.cpu().numpy().copy() for sc, p in ts.params.items()}
    losses += [net.train_step(sess, blobs, ts) for _ in range(steps - 1)]
    torch.cuda.synchronize()
    return losses, grads, calls


@pytest.mark.parametrize("min_tiles", [2, 150])
def test_fused_chain_rule_passes_change_nothing(dev, min_tiles):
    """One ResNet-50 training step with and without the folded passes (TrainState.fuse_chain).  min_tiles = 150 (no frcnn_gemm_h2 launch
    at this toy size, all float32): every filter gradient the sweep produces BEFORE its one order-dependent kernel (crop_and_resize's
    backward: float atomics) -- the tail and the heads -- is bit identical, the rest agrees to that kernel's noise (the bar of
    test_wgrad_side_streams_change_nothing).  min_tiles = 2: the frcnn_gemm_h2 data gradients run at this size too, some on dY planes
    out of the Winograd output transform (another valid scale than the splitter's: results differ in the last bits), so everything is
    held to the noise bar.  The fused sweep launches no residual copy, a fraction of the relu_bwd / h2_split passes, the same GEMMs."""
    from model.config import cfg
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_TRAIN_MIN_TILES = 64, 0.0, None      # None: TRAIN mode reads H2_MIN_TILES (set by _one_step)
    try:
        l0, g0, c0 = _one_step(dev, False, "cf0_%d" % min_tiles, min_tiles)
        l1, g1, c1 = _one_step(dev, True, "cf1_%d" % min_tiles, min_tiles)
        l2, g2, c2 = _one_step(dev, True, "cf2_%d" % min_tiles, min_tiles, pipe=False)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = old
    assert l0[0] == l1[0]
    assert np.allclose(np.array(l0[1:]), np.array(l1[1:]), rtol=1e-4, atol=0), (l0, l1)
    assert len(g0) == len(g1) > 40
    exact = 0
    for sc in g0:
        if min_tiles == 150 and ("block4" in sc or "cls_score" in sc or "bbox_pred" in sc):
            assert np.array_equal(g0[sc], g1[sc]), sc
            exact += 1
        else:
            assert np.abs(g0[sc] - g1[sc]).max() <= 1e-5 * max(np.abs(g0[sc]).max(), 1e-20), sc
    assert exact >= 11 or min_tiles != 150
    # TrainState.pipe_dgrads: the strided 3x3 and the odd-width 1x1 heads' data gradients as matrix-pipe convolutions of a spread-out /
    # zero-padded dY instead of the gather kernel -- another summation order, so every gradient behind them agrees to rounding only
    n = lambda c, k: c.get(k, 0)
    assert n(c2, "frcnn_conv2d_dgrad_strided") >= n(c1, "frcnn_conv2d_dgrad_strided") + 3
    for sc in g1:
        assert np.abs(g1[sc] - g2[sc]).max() <= 1e-5 * max(np.abs(g1[sc]).max(), 1e-20), sc
    assert n(c1, "frcnn_relu_bwd") <= n(c0, "frcnn_relu_bwd") // 3, (c0, c1)
    assert n(c1, "frcnn_h2_split") <= n(c0, "frcnn_h2_split")
    if min_tiles == 2:
        assert n(c1, "frcnn_gemm_h2_masked") > 10 and n(c1, "frcnn_h2_split") < n(c0, "frcnn_h2_split")
    assert n(c1, "frcnn_conv2d_nhwc_masked_ws") > 0 and n(c1, "frcnn_winograd_output_transform_masked") + n(c1, "frcnn_winograd7_output_transform_masked") > 10
    for k in ("frcnn_conv2d_wgrad_h2", "frcnn_conv2d_wgrad", "frcnn_gemm_batched_nt"):
        assert n(c0, k) == n(c1, k), k


@pytest.mark.parametrize("C", [256, 128, 64])
def test_winograd_row_per_thread_form_gives_the_bits_of_the_tile_per_thread_form(dev, C):
    """Launches of fewer than 256 workgroups (one 38 x 63 image: the training step, batch-1 inference) run F(4x4,3x3)'s transforms with one
    row of the tile per thread (csrc/winograd.hip k_wino4_input_rows / k_wino4_output_rows); a batch of 8 runs a tile per thread.  Same
    expressions -> the single image's result IS slot 3 of the batch's, bit for bit: V, planes of V, the output tensor, its planes, and
    the masked output of the reverse sweep."""
    from frcnn_hip import ops
    H, W, m, B = 38, 63, 4, 8
    G = ops.winograd_points(m)
    xb = _rand((B, H, W, C), dev, 21)
    x1 = xb[3:4].contiguous()
    T1, TB = ops.winograd_tiles(1, H, W, m), ops.winograd_tiles(B, H, W, m)
    v1 = ops.winograd_input_transform(x1, torch.empty((G, T1, C), device=dev), m)
    vb = ops.winograd_input_transform(xb, torch.empty((G, TB, C), device=dev), m)
    assert torch.equal(v1, vb[:, 3 * T1:4 * T1])
    if C % 128 == 0:
        p1 = ops.winograd_input_transform_h2(x1, ops.H2.empty(G * T1, C, dev), m)
        pb = ops.winograd_input_transform_h2(xb, ops.H2.empty(G * TB, C, dev), m)
        f1, fb = p1.to_float().view(G, T1, C), pb.to_float().view(G, TB, C)
        assert torch.equal(f1, fb[:, 3 * T1:4 * T1])
        assert torch.equal(p1.inv.view(C // 128, G, T1), pb.inv.view(C // 128, G, TB)[:, :, 3 * T1:4 * T1])
    mb = _rand((G, TB, C), dev, 22)
    m1 = mb[:, 3 * T1:4 * T1].contiguous()
    bias = _rand((C,), dev, 23)
    fwd = _rand((B, H, W, C), dev, 24)
    y1 = ops.winograd_output_transform(m1, bias, 1, torch.empty((1, H, W, C), device=dev), m)
    yb = ops.winograd_output_transform(mb, bias, 1, torch.empty((B, H, W, C), device=dev), m)
    assert torch.equal(y1[0], yb[3])
    k1 = ops.winograd_output_transform_masked(m1, fwd[3:4].contiguous(), torch.empty((1, H, W, C), device=dev), m)
    kb = ops.winograd_output_transform_masked(mb, fwd, torch.empty((B, H, W, C), device=dev), m)
    assert torch.equal(k1[0], kb[3])
    if C % 128 == 0:
        q1, qb = ops.H2.empty(H * W, C, dev), ops.H2.empty(B * H * W, C, dev)
        ops.winograd_output_transform_h2(m1, bias, 1, (1, H, W, C), m, q1)
        ops.winograd_output_transform_h2(mb, bias, 1, (B, H, W, C), m, qb)
        assert torch.equal(q1.to_float(), qb.to_float()[3 * H * W:4 * H * W])
        assert torch.equal(q1.inv, qb.inv[:, 3 * H * W:4 * H * W])
