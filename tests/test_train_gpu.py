"""GPU parity of the training-side kernels through the C ABI.
Deterministic parts vs the pinned oracle / the reference goldens; random subsampling vs its contract
(uniform k-subset of the candidates, counts as in the reference); losses vs torch float64 autograd."""
import numpy as np
import pytest
import torch

import frcnn_oracle as ora
import synth

pytestmark = pytest.mark.gpu
f32 = np.float32
IM_INFO = np.array([600, 1000, 1.6], dtype=f32)


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dtype is None else t.to(dev, dtype)


def _at_inputs(dev, golden):
    from frcnn_hip import ops
    gt = golden["targets"]["gt"]
    base = ops.generate_anchors(16)
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    return gt, base, anc


def test_anchor_target_layer_deterministic_part_vs_oracle_and_reference(dev, golden):
    from frcnn_hip import ops
    gt, base, anc = _at_inputs(dev, golden)
    lab, tg, iw, ow = [t.cpu().numpy() for t in ops.anchor_target_layer(T(gt, dev), 600, 1000, 38, 63, T(base, dev), seed=-1)]
    want = ora.anchor_target_layer(np.zeros((1, 38, 63, 18), dtype=f32), gt, IM_INFO, [16], anc, 9, rng=np.random.RandomState(0),
                                   batchsize=10 ** 9)                  # batch so large that nothing is subsampled
    assert np.array_equal(lab, want[0])                                # labels: bit-exact (f64 IoU, thresholds, gt-argmax ties)
    assert np.allclose(tg, want[1], rtol=0, atol=2e-6)                 # bbox_transform: device logf vs np.log
    assert np.array_equal(iw, want[2]) and np.array_equal(ow, want[3])
    # regression targets do not depend on the sampling: compare with the REFERENCE's own output too
    assert np.allclose(tg, golden["targets"]["at_targets"], rtol=0, atol=2e-6)
    assert (lab == 1).sum() > 0 and (lab == 0).sum() > 1000


def test_anchor_target_layer_subsampling_contract(dev, golden):
    from frcnn_hip import ops
    gt, base, anc = _at_inputs(dev, golden)
    full = ops.anchor_target_layer(T(gt, dev), 600, 1000, 38, 63, T(base, dev), seed=-1)[0].cpu().numpy().ravel()
    outs = []
    for seed in (3, 3, 4):
        lab, tg, iw, ow = [t.cpu().numpy() for t in ops.anchor_target_layer(T(gt, dev), 600, 1000, 38, 63, T(base, dev), seed=seed)]
        l = lab.ravel()
        nfg, nbg = int((l == 1).sum()), int((l == 0).sum())
        assert nfg == min(int((full == 1).sum()), 128) and nfg + nbg == 256        # anchor_target_layer.py:72-86
        assert np.all(full[l == 1] == 1) and np.all(full[l == 0] == 0)             # subsets of the candidates
        assert np.allclose(ow[ow > 0], 1.0 / 256) and int((ow > 0).sum()) == 4 * 256
        assert int((iw > 0).sum()) == 4 * nfg
        outs.append(l)
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])  # deterministic per seed
    # the reference's sampled output has the same counts
    ref = golden["targets"]["at_labels"].ravel()
    assert int((ref == 1).sum()) == int((outs[0] == 1).sum()) and int((ref == 0).sum()) == int((outs[0] == 0).sum())


def test_proposal_target_layer_contract_and_values(dev, golden):
    from frcnn_hip import ops
    gt = golden["targets"]["gt"]
    prob, dl = synth.rpn_outputs(38, 63, 9, seed=3)
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    rois_in, sc_in = ora.proposal_layer(prob, dl, IM_INFO, "TRAIN", [16], anc, 9)
    out = ops.proposal_target_layer(T(rois_in, dev), T(sc_in.ravel(), dev), T(gt, dev), 21, seed=5)
    rois, sc, labels, tg, iw, ow, counts = [t.cpu().numpy() for t in out]
    nfg_s, nbg_s, nfg_c, nbg_c = counts.tolist()
    ov = ora.bbox_overlaps(rois_in[:, 1:5], gt[:, :4])
    mx, am = ov.max(axis=1), ov.argmax(axis=1)
    assert nfg_c == int((mx >= 0.5).sum()) and nbg_c == int(((mx < 0.5) & (mx >= 0.0)).sum())
    assert nfg_s == min(64, nfg_c) and nfg_s + nbg_s == 256                              # proposal_target_layer.py:119-127
    # every output row is one of the input rois, fg rows first
    idx = [int(np.where((rois_in == r).all(axis=1))[0][0]) for r in rois]
    assert np.all(mx[idx[:nfg_s]] >= 0.5) and np.all(mx[idx[nfg_s:]] < 0.5)
    assert len(set(idx[:nfg_s])) == nfg_s                                                # fg without replacement
    assert np.array_equal(sc, sc_in.ravel()[idx])
    want_labels = gt[am[idx], 4].copy()
    want_labels[nfg_s:] = 0
    assert np.array_equal(labels.ravel(), want_labels)
    # regression targets / weights: the oracle's formulas on the rows the kernel picked
    t = ora.bbox_transform(rois[:, 1:5], gt[am[idx], :4])
    t = ((t - np.array((0.0, 0.0, 0.0, 0.0))) / np.array((0.1, 0.1, 0.2, 0.2))).astype(f32)
    want_t = np.zeros((256, 84), dtype=f32)
    want_i = np.zeros((256, 84), dtype=f32)
    for i in np.where(want_labels > 0)[0]:
        c = int(4 * want_labels[i])
        want_t[i, c:c + 4] = t[i]
        want_i[i, c:c + 4] = 1
    assert np.allclose(tg, want_t, rtol=0, atol=2e-5) and np.array_equal(iw, want_i) and np.array_equal(ow, want_i)
    # same seed -> same sample; the reference's own sample has the same fg/bg split
    again = ops.proposal_target_layer(T(rois_in, dev), T(sc_in.ravel(), dev), T(gt, dev), 21, seed=5)[0].cpu().numpy()
    assert np.array_equal(again, rois)
    assert int((golden["targets"]["pt_labels"] > 0).sum()) == nfg_s


def test_losses_value_and_gradient_vs_torch_float64(dev, golden):
    from frcnn_hip import ops
    rng = np.random.RandomState(0)
    A, H, W = 9, 38, 63
    score = rng.randn(1, H, W, 2 * A).astype(f32)
    labels = golden["targets"]["at_labels"]                                           # [1,1,A*H,W] in {-1,0,1}
    loss, grad = ops.softmax_ce_loss(T(score, dev), T(labels, dev), rpn_shape=(A, H, W))
    s = torch.from_numpy(score).double().requires_grad_(True)
    # network.py:68-78,282-288: pair (a, A+a), element order (a, h, w)
    pair = torch.stack([s[0, :, :, :A].permute(2, 0, 1).reshape(-1), s[0, :, :, A:].permute(2, 0, 1).reshape(-1)], dim=1)
    lab = torch.from_numpy(labels.ravel()).long()
    sel = lab >= 0
    ref = torch.nn.functional.cross_entropy(pair[sel], lab[sel])
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 and np.allclose(grad.cpu().numpy(), s.grad.numpy(), rtol=0, atol=1e-8)
    # RCNN class loss (network.py:299-301)
    cs = (rng.randn(256, 21) * 2).astype(f32)
    lb = golden["targets"]["pt_labels"].ravel().astype(f32)
    loss, grad = ops.softmax_ce_loss(T(cs, dev), T(lb, dev))
    c = torch.from_numpy(cs).double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(c, torch.from_numpy(lb).long())
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 and np.allclose(grad.cpu().numpy(), c.grad.numpy(), rtol=0, atol=1e-8)
    # SmoothL1 (network.py:264-277): RPN sigma 3 / sum over (H,W,4A) / mean over batch 1; RCNN sigma 1 / mean over 256
    for pred_shape, tg, iw, ow, sigma, div in (
            ((1, H, W, 4 * A), golden["targets"]["at_targets"], golden["targets"]["at_inside"], golden["targets"]["at_outside"], 3.0, 1.0),
            ((256, 84), golden["targets"]["pt_targets"], golden["targets"]["pt_inside"], golden["targets"]["pt_outside"], 1.0, 256.0)):
        pred = (rng.randn(*pred_shape) * 0.5).astype(f32)
        loss, grad = ops.smooth_l1_loss(T(pred, dev), T(tg, dev), T(iw, dev), T(ow, dev), sigma, div)
        p = torch.from_numpy(pred).double().requires_grad_(True)
        d = torch.from_numpy(iw).double() * (p - torch.from_numpy(tg).double())
        s2 = sigma ** 2
        f = torch.where(d.abs() < 1.0 / s2, d * d * (s2 / 2.0), d.abs() - 0.5 / s2)
        ref = (torch.from_numpy(ow).double() * f).sum() / div
        ref.backward()
        assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
        assert np.allclose(grad.cpu().numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("wino,h2", [(False, False), (True, False), (True, True)], ids=["direct", "winograd", "winograd_h2"])
def test_full_train_step_matches_torch_autograd(dev, wino, h2):
    """TRAIN forward + reverse sweep + momentum SGD on the device vs torch float64
    autograd of the reference graph, with the device-sampled rois/targets fed to the reference as constants.
    wino: the 3x3 stride-1 layers' forward and data gradient as Winograd F(4x4,3x3) with device-transformed filters
    (cfg.HIP.WINOGRAD_TRAIN, the default) or on the direct kernels -- same bounds.  h2: cfg.HIP.H2_TRAIN forced onto this small
    network (H2_MIN_TILES = 1): the pointwise convolutions of the forward pass and their data gradients run in frcnn_gemm_h2
    (filters re-split after the solver step) -- same bounds again."""
    from dense_ref import TrainRef
    from frcnn_hip import ops as ops_mod
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    SC, RT = (4, 8, 16), (0.5, 1, 2)
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.WINOGRAD_TRAIN)
    old_h2 = (cfg.HIP.H2_TRAIN, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.WINOGRAD_TRAIN = 64, 0.0, wino
    cfg.HIP.H2_TRAIN, cfg.HIP.H2_MIN_TILES = bool(h2), (1 if h2 else cfg.HIP.H2_MIN_TILES)
    cfg.HIP.H2_TRAIN_MIN_TILES = None if h2 else cfg.HIP.H2_TRAIN_MIN_TILES       # None: TRAIN mode reads the same knob as TEST mode
    try:
        sess = Session(device=dev, seed=5)
        net = resnetv1(num_layers=50)
        net.create_architecture("TRAIN", 21, tag="train_w%d_h%d" % (int(wino), int(h2)), anchor_scales=SC, anchor_ratios=RT)
        sess.init_variables(net.variable_specs())
        rng = np.random.RandomState(2)
        H, W = 128, 160
        image = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
        gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)
        blobs = dict(data=image, im_info=np.array([H, W, 1.0], dtype=np.float32), gt_boxes=gt)
        losses = net.train_forward(sess, blobs)
        pt, at = net._proposal_targets, net._anchor_targets
        counts = pt["counts"].cpu().numpy()
        assert counts[0] > 0 and counts[0] + counts[1] == 64
        ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4).build()
        ts.winograd = (4, 64, True) if wino else None
        ts.h2_train = 1 if h2 else None
        # filter gradients: "direct" keeps round 2's transposes + forward GEMM kernel, "winograd" the f32 TN kernel, "winograd_h2" the
        # fp16-pipe TN kernel (csrc/wgrad_tn.hip, wgrad_h2.hip); the latter two on two side streams
        ts.wgrad_tn, ts.wgrad_h2, ts.wgrad_stream = wino, h2, (2 if wino else 0)
        if h2:
            assert len(sess.h2) >= 20, "the forward pass did not take the h2 path"
        ts.backward(net._loss_seeds)
        torch.cuda.synchronize()
        ref = TrainRef(sess.variables, 50, 21, SC, RT, net.trainable_scope)
        to_np = lambda d_: {k: v.cpu().numpy() for k, v in d_.items()}
        rl = ref.losses(image, pt["rois"].cpu().numpy(), to_np(at), to_np(pt))
        for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box"):
            assert abs(losses[k].item() - rl[k].item()) <= 1e-4 * max(1.0, abs(rl[k].item())), k
        sum(rl.values()).backward()
        checked = 0
        for sc in ("/cls_score", "/bbox_pred", "/rpn_cls_score", "/rpn_conv/3x3", "/block4/unit_3/bottleneck_v1/conv2",
                   "/block4/unit_1/bottleneck_v1/shortcut", "/block3/unit_6/bottleneck_v1/conv3", "/block3/unit_1/bottleneck_v1/conv1",
                   "/block2/unit_4/bottleneck_v1/conv2", "/block2/unit_1/bottleneck_v1/conv1", "/block2/unit_1/bottleneck_v1/shortcut"):
            scope = net._scope + sc
            p = ts.params[scope]
            g = ref._cache[scope + "/weights"].grad.numpy()                        # HWIO (or [in,out])
            if g.ndim == 2:
                g = g[None, None]
            want = np.transpose(g, (3, 0, 1, 2))                                    # -> [Cout,KH,KW,Cin] (master filter)
            got = p.grad_w.cpu().numpy()
            if p.scale is not None:
                got = got * p.scale.cpu().numpy()[:, None, None, None]              # chain rule through the BN fold
            err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-12)
            assert err <= 2e-3, (sc, err)
            if p.bias is not None:
                gb = ref._cache[scope + "/biases"].grad.numpy()
                assert np.abs(p.grad_b.cpu().numpy() - gb).max() <= 2e-3 * max(np.abs(gb).max(), 1e-12), sc
            checked += 1
        assert checked == 11 and net._scope + "/conv1" not in ts.params and not any("/block1/" in s for s in ts.params)
        # one solver step (train_val.py:128-145): acc = g + wd*w ; w -= lr*acc ; folded copy refreshed
        sc = net._scope + "/block3/unit_1/bottleneck_v1/conv1"
        p = ts.params[sc]
        w0, g0 = p.w.cpu().numpy().copy(), (p.grad_w * p.scale.view(-1, 1, 1, 1)).cpu().numpy()
        ts.apply(lr=0.01)
        w1 = p.w.cpu().numpy()
        assert np.allclose(w1, w0 - 0.01 * (g0 + 1e-4 * w0), rtol=1e-5, atol=1e-7)
        assert np.allclose(p.wf.cpu().numpy(), w1 * p.scale.cpu().numpy()[:, None, None, None], rtol=1e-6, atol=1e-8)
        # and the public API: a second full step runs and returns the five losses (network.py:488-498)
        ts.lr = 0.001
        out = net.train_step(sess, blobs, ts)
        assert len(out) == 5 and all(np.isfinite(out)) and out[4] > sum(out[:4])   # total includes the L2 term
        if h2:                                       # the filter planes follow the solver: re-split in place after apply()
            wq, wsrc = next(iter(sess.h2.values()))
            fresh = ops_mod.h2_pack_w(wsrc)
            assert torch.equal(fresh[0], wq[0]) and torch.equal(fresh[1], wq[1])
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.WINOGRAD_TRAIN = old
        cfg.HIP.H2_TRAIN, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = old_h2


def test_sgd_steps_on_a_fixed_batch_reduce_the_loss(dev):
    """Sanity of the whole loop (forward, reverse sweep, momentum SGD, BN-fold refresh): repeated steps on one
    fixed image + gt with a fixed sampling seed must drive the four task losses down."""
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    try:
        sess = Session(device=dev, seed=5)
        net = resnetv1(num_layers=50)
        net.create_architecture("TRAIN", 21, tag="train2", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess.init_variables(net.variable_specs())
        rng = np.random.RandomState(2)
        image = ((rng.rand(1, 128, 160, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
        gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)
        blobs = dict(data=image, im_info=np.array([128, 160, 1.0], dtype=np.float32), gt_boxes=gt)
        ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4)
        ts.lr = 2e-4
        hist = []
        for _ in range(24):
            net._sample_seed = 0                          # same fg/bg sample every step
            out = net.train_step(sess, blobs, ts)
            hist.append(sum(out[:4]))
        assert all(np.isfinite(hist))
        assert np.mean(hist[-4:]) < 0.8 * np.mean(hist[:4]), hist
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = old


@pytest.mark.gpu
def test_wgrad_side_streams_change_nothing(dev):
    """cfg.HIP.WGRAD_STREAM: the filter gradients run on side streams beside the data-gradient chain.  Same kernels on the same operands
    -> the momentum slots after the first step (= the gradients) and the losses of four steps agree with the one-stream run to the noise
    of the one order-dependent kernel of the sweep (the float atomics of crop_and_resize's backward).  Also under the data-parallel rules
    with one replica (TrainState.force_dp: at most one side stream)."""
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.WGRAD_STREAM)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    rng = np.random.RandomState(4)
    image = ((rng.rand(1, 160, 224, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [100, 70, 200, 150, 12]], dtype=np.float32)
    blobs = dict(data=image, im_info=np.array([160, 224, 1.0], dtype=np.float32), gt_boxes=gt)
    try:
        state = []
        for n, (side, dp) in enumerate([(0, False), (2, False), (2, True)]):
            cfg.HIP.WGRAD_STREAM = side
            sess = Session(device=dev, seed=9)
            net = resnetv1(num_layers=50)
            net.create_architecture("TRAIN", 21, tag="ws%d" % n, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
            sess.init_variables(net.variable_specs())
            ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4)
            ts.lr = 1e-3
            ts.force_dp = dp
            losses = [net.train_step(sess, blobs, ts)]
            torch.cuda.synchronize()
            slots = {sc: (p.acc_w.cpu().numpy().copy(), None if p.acc_b is None else p.acc_b.cpu().numpy().copy()) for sc, p in ts.params.items()}
            losses += [net.train_step(sess, blobs, ts) for _ in range(3)]
            torch.cuda.synchronize()
            assert len(getattr(ts, "_wgrad_stream_objs", [])) == (min(side, 1) if dp else side)
            state.append((losses, slots))
        l0, p0 = state[0]
        for l1, p1 in state[1:]:
            assert l0[0] == l1[0], (l0, l1)                               # the first forward pass: no order-dependent kernel
            assert np.allclose(np.array(l0[1:]), np.array(l1[1:]), rtol=1e-4, atol=0), (l0, l1)
            assert len(p0) == len(p1) > 40
            for sc in p0:
                for a, b in zip(p0[sc], p1[sc]):
                    assert (a is None and b is None) or np.abs(a - b).max() <= 1e-5 * max(np.abs(a).max(), 1e-20), sc
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.WGRAD_STREAM = old


def test_snapshot_and_resume_continue_the_same_run(dev, tmp_path):
    """train_val.py:58-100,204-233 on TF V2 bundles written without TensorFlow: run 4 steps; run 2 steps, snapshot, restore
    into a fresh session + solver (weights, Momentum slots, iteration, sampling stream), run 2 more -> the same losses as
    the uninterrupted run.  The restored solver state (weights, Momentum, iteration, sampling seed) is compared BIT FOR BIT with
    the state the interrupted run had; loss trajectories of separate runs are only compared loosely (float atomics)."""
    from frcnn_hip.runtime import Session
    from frcnn_hip.tensor_bundle import BundleReader
    from model.config import cfg
    from model.train_val import SolverWrapper
    from nets.resnet_v1 import resnetv1
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.STEPSIZE, cfg.TRAIN.DISPLAY)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.STEPSIZE = 64, 0.0, 2e-4, 2, [3]
    cfg.TRAIN.DISPLAY = 1000
    rng = np.random.RandomState(2)
    image = ((rng.rand(1, 128, 160, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)

    def layer():
        while True:
            yield dict(data=image, im_info=np.array([128, 160, 1.0], dtype=np.float32), gt_boxes=gt)

    def solver(tag):
        sess = Session(device=dev, seed=5)
        net = resnetv1(num_layers=50)
        net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess.init_variables(net.variable_specs())
        return sess, net, SolverWrapper(sess, net, layer())

    try:
        _, _, sw = solver("snapA")
        full = sw.train_model(4, verbose=False)
        _, _, sw = solver("snapB")
        first = sw.train_model(2, verbose=False, snapshot_dir=str(tmp_path))
        at_snapshot = sw.state.export_variables(slots=True)                         # weights + Momentum after iteration 2
        ck, pk = str(tmp_path / "res101_faster_rcnn_iter_2.ckpt"), str(tmp_path / "res101_faster_rcnn_iter_2.pkl")
        shapes = BundleReader(ck).get_variable_to_shape_map()
        assert shapes["global_step"] == [] and "resnet_v1_50/block3/unit_1/bottleneck_v1/conv2/weights/Momentum" in shapes
        assert "resnet_v1_50/conv1/weights/Momentum" not in shapes                     # frozen stem: no optimizer slot
        sess, net, sw = solver("snapC")
        assert sw.restore(ck, pk) == 2 and net._sample_seed == 4                       # iteration + sampling stream
        # the restored solver state, before any update: build the parameter set exactly as the first resumed step does
        net.train_forward(sess, next(layer()))
        sw.state.build()
        sw.state.import_slots(sw.state.pending_slots)
        sw.state.pending_slots = None
        restored = sw.state.export_variables(slots=True)
        assert sorted(restored) == sorted(at_snapshot)
        for k in at_snapshot:                                                          # bit-identical weights and momentum
            assert np.array_equal(restored[k], at_snapshot[k]), k
        rest = sw.train_model(4, verbose=False, start_iter=2)                          # continues at iteration 3 with the decayed lr at 4
        # trajectories of two separate runs agree only loosely: the crop backward adds with float atomics, and a 1e-7 change
        # of the weights can flip an NMS / sampling decision downstream
        assert np.allclose(first, full[:2], rtol=1e-3, atol=0) and np.allclose(rest, full[2:], rtol=5e-2, atol=0), (first + rest, full)
    finally:
        (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.STEPSIZE, cfg.TRAIN.DISPLAY) = old


def _train_once(dev, net, tag, batch=64):
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    SC, RT = (4, 8, 16), (0.5, 1, 2)
    sess = Session(device=dev, seed=5)
    net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=SC, anchor_ratios=RT)
    sess.init_variables(net.variable_specs())
    rng = np.random.RandomState(2)
    H, W = 128, 160
    image = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)
    blobs = dict(data=image, im_info=np.array([H, W, 1.0], dtype=np.float32), gt_boxes=gt)
    losses = net.train_forward(sess, blobs)
    ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4).build()
    ts.winograd = None
    ts.backward(net._loss_seeds)
    torch.cuda.synchronize()
    return sess, ts, blobs, image, losses, (SC, RT)


def _check_param_grads(net, ts, ref, scopes, tol=2e-3):
    for sc in scopes:
        scope = net._scope + sc
        p = ts.params[scope]
        if getattr(p, "dw", False):
            want = ref._cache[scope + "/depthwise_weights"].grad.numpy()[:, :, :, 0]       # [3,3,C]
            got = p.grad_w.cpu().numpy()                                                    # chain rule through the fold already applied
        else:
            g = ref._cache[scope + "/weights"].grad.numpy()
            if g.ndim == 2:
                g = g[None, None]
            want = np.transpose(g, (3, 0, 1, 2))
            got = p.grad_w.cpu().numpy()
            if p.scale is not None:
                got = got * p.scale.cpu().numpy()[:, None, None, None]
        assert np.abs(want).max() > 0, sc
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= tol, (sc, err)
        if p.bias is not None:
            gb = ref._cache[scope + "/biases"].grad.numpy()
            assert np.abs(p.grad_b.cpu().numpy() - gb).max() <= tol * max(np.abs(gb).max(), 1e-12), sc


def test_vgg16_train_step_matches_torch_autograd(dev):
    """VGG16 TRAIN graph (vgg16.py:26-60): max pools, 14x14 crop + 2x2 max, fc6 / fc7 with dropout -- forward losses and
    parameter gradients vs torch float64 autograd; the device's dropout masks and sampled rois / targets are the reference's constants."""
    from dense_ref import VGG16TrainRef
    from frcnn_hip import ops
    from model.config import cfg
    from nets.vgg16 import vgg16
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    try:
        net = vgg16()
        sess, ts, blobs, image, losses, (SC, RT) = _train_once(dev, net, "train_vgg")
        pt, at = net._proposal_targets, net._anchor_targets
        drops = [r for r in net._tape if r["kind"] == "dropout"]
        assert len(drops) == 2
        masks = []
        for r in drops:
            m = ops.dropout(torch.ones_like(r["x"]), r["seed"], r["keep"]).view(-1, 4096).cpu().numpy()
            assert set(np.unique(m)) <= {0.0, 2.0} and 0.45 < (m > 0).mean() < 0.55
            got = r["y"].view(-1, 4096).cpu().numpy()
            assert np.array_equal(got, r["x"].view(-1, 4096).cpu().numpy() / np.float32(0.5) * (m > 0))
            masks.append(m)
        assert not np.array_equal(masks[0], masks[1])
        ref = VGG16TrainRef(sess.variables, 21, SC, RT, net.trainable_scope, masks)
        to_np = lambda d_: {k: v.cpu().numpy() for k, v in d_.items()}
        rl = ref.losses(image, pt["rois"].cpu().numpy(), to_np(at), to_np(pt))
        for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box"):
            assert abs(losses[k].item() - rl[k].item()) <= 1e-4 * max(1.0, abs(rl[k].item())), k
        sum(rl.values()).backward()
        _check_param_grads(net, ts, ref, ("/cls_score", "/bbox_pred", "/rpn_cls_score", "/rpn_bbox_pred", "/rpn_conv/3x3", "/fc7", "/fc6",
                                          "/conv5/conv5_3", "/conv5/conv5_1", "/conv4/conv4_3", "/conv4/conv4_1", "/conv3/conv3_2", "/conv3/conv3_1"))
        assert not any("/conv1/" in s or "/conv2/" in s for s in ts.params)
        ts.lr = 0.001
        out = net.train_step(sess, blobs, ts)
        assert len(out) == 5 and all(np.isfinite(out)) and out[4] > sum(out[:4])
        net.train_forward(sess, blobs)                                                                      # the forward of the NEXT step
        assert [r["seed"] for r in net._tape if r["kind"] == "dropout"] != [r["seed"] for r in drops]      # a new mask every step
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = old


def test_mobilenet_train_step_matches_torch_autograd(dev):
    """MobileNet-v1 TRAIN graph (mobilenet_v1.py:214-250): depthwise / pointwise layers >= FIXED_LAYERS train (frozen BN folded,
    ReLU6), 14x14 crop + 2x2 max, two per-RoI layers + mean -- losses, pointwise and depthwise filter gradients vs float64 autograd,
    one solver step on a depthwise filter."""
    from dense_ref import MobileNetTrainRef
    from model.config import cfg
    from nets.mobilenet_v1 import mobilenetv1
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    try:
        net = mobilenetv1()
        sess, ts, blobs, image, losses, (SC, RT) = _train_once(dev, net, "train_mob")
        pt, at = net._proposal_targets, net._anchor_targets
        ref = MobileNetTrainRef(sess.variables, 21, SC, RT, net.trainable_scope)
        to_np = lambda d_: {k: v.cpu().numpy() for k, v in d_.items()}
        rl = ref.losses(image, pt["rois"].cpu().numpy(), to_np(at), to_np(pt))
        for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box"):
            assert abs(losses[k].item() - rl[k].item()) <= 1e-4 * max(1.0, abs(rl[k].item())), k
        sum(rl.values()).backward()
        _check_param_grads(net, ts, ref, ("/cls_score", "/bbox_pred", "/rpn_cls_score", "/rpn_conv/3x3",
                                          "/Conv2d_13_pointwise", "/Conv2d_13_depthwise", "/Conv2d_12_pointwise", "/Conv2d_12_depthwise",
                                          "/Conv2d_11_pointwise", "/Conv2d_11_depthwise", "/Conv2d_6_pointwise", "/Conv2d_6_depthwise",
                                          "/Conv2d_5_pointwise", "/Conv2d_5_depthwise"))
        fixed = ["/Conv2d_%d_" % i for i in range(0, cfg.MOBILENET.FIXED_LAYERS)]
        assert not any(f in s for s in ts.params for f in fixed) and net._scope + "/Conv2d_0" not in ts.params
        # solver step on a depthwise filter: no L2 term (MOBILENET.REGU_DEPTH False), folded copy refreshed
        p = ts.params[net._scope + "/Conv2d_6_depthwise"]
        w0, g0 = p.w.cpu().numpy().copy(), p.grad_w.cpu().numpy().copy()
        q = ts.params[net._scope + "/Conv2d_6_pointwise"]
        qw0, qg0 = q.w.cpu().numpy().copy(), (q.grad_w * q.scale.view(-1, 1, 1, 1)).cpu().numpy()
        ts.apply(lr=0.01)
        w1 = p.w.cpu().numpy()
        assert np.allclose(w1, w0 - 0.01 * g0, rtol=1e-5, atol=1e-8)
        assert np.allclose(p.wf.cpu().numpy(), w1 * p.scale.cpu().numpy()[None, None, :], rtol=1e-6, atol=1e-9)
        assert np.allclose(q.w.cpu().numpy(), qw0 - 0.01 * (qg0 + cfg.MOBILENET.WEIGHT_DECAY * qw0), rtol=1e-5, atol=1e-8)   # backbone coefficient
        ts.lr = 0.001
        out = net.train_step(sess, blobs, ts)
        assert len(out) == 5 and all(np.isfinite(out)) and out[4] > sum(out[:4])
        exp = ts.export_variables(slots=True)
        name = net._scope + "/Conv2d_6_depthwise/depthwise_weights"
        assert exp[name].shape == (3, 3, p.w.shape[-1], 1) and exp[name + "/Momentum"].shape == exp[name].shape
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = old


def test_maxpool_and_depthwise_gradients_vs_torch(dev):
    """frcnn_maxpool_bwd (2x2/2 'SAME' on odd sizes; 3x3/2 overlapping windows) and the depthwise data / filter gradients
    (stride 1 and 2) against torch float64 autograd."""
    from frcnn_hip import ops
    rng = np.random.RandomState(4)
    for (H, W, k, s) in ((37, 25, 2, 2), (14, 14, 2, 2), (19, 23, 3, 2)):
        x = rng.randn(2, H, W, 8).astype(np.float32)
        x[0, :4, :4] = 0.5                                                        # ties: the first maximum takes the gradient
        OH, OW = -(-(H - (k - s)) // s) if k > s else -(-H // s), -(-(W - (k - s)) // s) if k > s else -(-W // s)
        xt = torch.from_numpy(x).double().permute(0, 3, 1, 2).requires_grad_(True)
        if k == s:
            yt = torch.nn.functional.max_pool2d(xt, k, s, ceil_mode=True)
        else:
            yt = torch.nn.functional.max_pool2d(xt, k, s)
        OH, OW = yt.shape[2], yt.shape[3]
        g = rng.randn(2, OH, OW, 8).astype(np.float32)
        yt.backward(torch.from_numpy(g).double().permute(0, 3, 1, 2))
        xd = torch.from_numpy(x).to(dev)
        pad = (0, (OH - 1) * s + k - H, 0, (OW - 1) * s + k - W)
        yd = ops.maxpool(xd, k, s, pad)
        assert np.array_equal(yd.cpu().numpy(), yt.detach().permute(0, 2, 3, 1).numpy().astype(np.float32))
        dx = ops.maxpool_bwd(xd, yd, torch.from_numpy(g).to(dev), k, s, torch.empty_like(xd))
        assert np.allclose(dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-6, atol=1e-6), (H, W, k, s)
    for stride in (1, 2):
        N, H, W, C = 2, 17, 21, 32
        x = rng.randn(N, H, W, C).astype(np.float32)
        w = rng.randn(3, 3, C).astype(np.float32)
        scale = (rng.rand(C) + 0.5).astype(np.float32)
        xt = torch.from_numpy(x).double().permute(0, 3, 1, 2).requires_grad_(True)
        wt = torch.from_numpy(w).double().requires_grad_(True)                                      # master filter
        wf = (wt * torch.from_numpy(scale).double()[None, None, :]).permute(2, 0, 1)[:, None]       # folded, [C,1,3,3]
        yt = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (1, 1, 1, 1)), wf, stride=stride, groups=C)
        g = rng.randn(N, yt.shape[2], yt.shape[3], C).astype(np.float32)
        yt.backward(torch.from_numpy(g).double().permute(0, 3, 1, 2))
        gd, xd = torch.from_numpy(g).to(dev), torch.from_numpy(x).to(dev)
        wfd = torch.from_numpy(w * scale[None, None, :]).to(dev)
        dx = ops.dwconv3x3_dgrad(gd, wfd, stride, (1, 1, 1, 1), torch.empty_like(xd))
        assert np.allclose(dx.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-5, atol=1e-5)
        dw = ops.dwconv3x3_wgrad(gd, xd, stride, (1, 1, 1, 1), torch.from_numpy(scale).to(dev), torch.empty((3, 3, C), device=dev))
        want = wt.grad.numpy()
        assert np.abs(dw.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("arch", ["mobile", "vgg16"])
def test_snapshot_restore_is_bit_identical_for_mobilenet_and_vgg16(dev, tmp_path, arch):
    """The solver state of the other two backbone families survives a snapshot: depthwise master filters
    (`.../depthwise_weights` [3,3,C,1]) and their Momentum slots for MobileNet, fc6 / fc7 matrices and biases for VGG16."""
    from frcnn_hip.runtime import Session
    from frcnn_hip.tensor_bundle import BundleReader
    from model.config import cfg
    from model.train_val import SolverWrapper
    from nets.mobilenet_v1 import mobilenetv1
    from nets.vgg16 import vgg16
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.DISPLAY)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.DISPLAY = 64, 0.0, 2e-4, 2, 1000
    rng = np.random.RandomState(2)
    image = ((rng.rand(1, 128, 160, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)

    def layer():
        while True:
            yield dict(data=image, im_info=np.array([128, 160, 1.0], dtype=np.float32), gt_boxes=gt)

    def solver(tag):
        sess = Session(device=dev, seed=5)
        net = mobilenetv1() if arch == "mobile" else vgg16()
        net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess.init_variables(net.variable_specs())
        return sess, net, SolverWrapper(sess, net, layer())

    try:
        _, net, sw = solver(arch + "_snapA")
        losses = sw.train_model(2, verbose=False, snapshot_dir=str(tmp_path))
        assert len(losses) == 2 and np.all(np.isfinite(losses))
        at_snapshot = sw.state.export_variables(slots=True)
        import glob
        ck = sorted(glob.glob(str(tmp_path / "*_iter_2.ckpt*")))[0].split(".ckpt")[0] + ".ckpt"
        pk = ck[:-5] + ".pkl"
        shapes = BundleReader(ck).get_variable_to_shape_map()
        probe = (net._scope + "/Conv2d_6_depthwise/depthwise_weights") if arch == "mobile" else (net._scope + "/fc6/weights")
        assert probe in shapes and probe + "/Momentum" in shapes
        assert shapes[probe] == ([3, 3, 256, 1] if arch == "mobile" else [7 * 7 * 512, 4096])
        sess, net2, sw2 = solver(arch + "_snapB")
        assert sw2.restore(ck, pk) == 2
        net2.train_forward(sess, next(layer()))
        sw2.state.build()
        sw2.state.import_slots(sw2.state.pending_slots)
        sw2.state.pending_slots = None
        restored = sw2.state.export_variables(slots=True)
        assert sorted(restored) == sorted(at_snapshot)
        for k in at_snapshot:
            assert np.array_equal(restored[k], at_snapshot[k]), k
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.DISPLAY = old
