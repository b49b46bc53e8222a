"""The training step as a recorded launch list (cfg.HIP.TRAIN_REPLAY, frcnn_hip/replay.py; the reference's step is ONE sess.run of a
graph built once, lib/nets/network.py:488-498): the second step of an image shape is recorded while it runs, later steps replay the list.
A replayed step makes the launches of the eager step with the same arguments on the same streams in the same order, and since round 5
no kernel of the step adds in a data-dependent order (crop_and_resize's backward is a gather), so EVERYTHING must agree bit for bit with
a run that enqueues every step from Python: the five losses of every step, every parameter and every momentum accumulator at the end --
for all three backbone families, with images whose ground-truth box count changes from step to step (a patched launch argument), with
the sampling seeds advancing (patched), and with dropout masks advancing (VGG16, patched)."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobs(H, W, n=3, seed=4):
    from model.config import cfg
    rng = np.random.RandomState(seed)
    out = []
    gts = [np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [100, 70, 200, 150, 12]], dtype=np.float32),
           np.array([[20, 24, 120, 140, 5]], dtype=np.float32),
           np.array([[8, 8, 60, 90, 1], [90, 20, 210, 100, 2], [30, 100, 140, 200, 9], [150, 110, 270, 210, 15], [5, 120, 70, 215, 18]], dtype=np.float32)]
    for i in range(n):
        image = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
        out.append(dict(data=image, im_info=np.array([H, W, 1.0], dtype=np.float32), gt_boxes=gts[i % len(gts)]))
    return out


def _digest(ts):
    h = hashlib.sha256()
    for sc in sorted(ts.params):
        p = ts.params[sc]
        for t in (p.w, p.acc_w, p.bias, p.acc_b):
            if t is not None:
                h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def _run(dev, make_net, tag, replay, steps, H=224, W=288, classes=21):
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    old = cfg.HIP.TRAIN_REPLAY
    cfg.HIP.TRAIN_REPLAY = replay
    try:
        sess = Session(device=dev, seed=9)
        net = make_net()
        net.create_architecture("TRAIN", classes, tag=tag, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess.init_variables(net.variable_specs())
        ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4)
        ts.lr = 1e-3
        blobs = _blobs(H, W)
        losses = [net.train_step(sess, blobs[i % len(blobs)], ts) for i in range(steps)]
        torch.cuda.synchronize()
        return losses, _digest(ts), dict(net.replay_stats), net, sess, ts
    finally:
        cfg.HIP.TRAIN_REPLAY = old


def _nets():
    from nets.mobilenet_v1 import mobilenetv1
    from nets.resnet_v1 import resnetv1
    from nets.vgg16 import vgg16
    return {"res50": lambda: resnetv1(num_layers=50), "vgg16": vgg16, "mobile": mobilenetv1}


@pytest.mark.parametrize("family", ["res50", "vgg16", "mobile"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(dev, family):
    from model.config import cfg
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = 64, 0.0, 2, None   # (frcnn_gemm_h2 launches at this toy size too)
    try:
        make = _nets()[family]
        l0, d0, s0, _, _, _ = _run(dev, make, "rp0_" + family, False, 7)
        l1, d1, s1, net, _, _ = _run(dev, make, "rp1_" + family, True, 7)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = old
    assert s0 == dict(eager=0, recorded=0, replayed=0)
    assert s1 == dict(eager=1, recorded=1, replayed=5), s1       # step 1 eager (builds everything), step 2 recorded, steps 3-7 replayed
    assert all(np.isfinite(v) for step in l0 for v in step)
    assert len({tuple(s) for s in l0}) == 7                       # seven different steps (images, seeds, weights move)
    assert l1 == l0, (l0, l1)                                      # float -> python float is exact: the same bits
    assert d1 == d0
    # a replayed step leaves the network's views of its outputs in place (the tensors are static)
    assert net._predictions["cls_score"].shape[0] == 64 and set(net._losses) == {"cross_entropy", "loss_box", "rpn_cross_entropy", "rpn_loss_box"}


def _save_state(ts):
    return {sc: tuple(None if t is None else t.clone() for t in (p.w, p.acc_w, p.wf, p.bias, p.acc_b)) for sc, p in ts.params.items()}


def _load_state(ts, net, state, seed):
    for sc, saved in state.items():
        p = ts.params[sc]
        for t, v in zip((p.w, p.acc_w, p.wf, p.bias, p.acc_b), saved):
            if v is not None and t is not None:
                t.copy_(v)
    ts.refresh_derived()                       # operand planes / Winograd images / prepared gradient filters of the restored filters
    net._sample_seed = seed


def _repeat_one_step(net, sess, ts, blob, repeats, start=None):
    """the SAME step (state, image, seeds) `repeats` times: -> set of (loss bits, parameter digest); start = (state, seed) to begin from
    (default: where the solver is now)"""
    state, seed = (_save_state(ts), net._sample_seed) if start is None else start
    seen = set()
    for _ in range(repeats):
        _load_state(ts, net, state, seed)
        out = net.train_step(sess, blob, ts)
        torch.cuda.synchronize()
        seen.add((tuple(np.float32(v).tobytes() for v in out), _digest(ts)))
    return seen


@pytest.mark.parametrize("replay", [False, True])
def test_twenty_repeats_of_one_step_are_bit_identical(dev, replay):
    """Determinism of the step itself (what makes the comparison above meaningful): the same state, the same image, the same seeds ->
    the same loss bits and the same parameter / momentum digest, twenty times over, eagerly and replayed (crop_and_resize's backward was
    the last kernel that added in hardware order)."""
    from model.config import cfg
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    try:
        _, _, _, net, sess, ts = _run(dev, _nets()["res50"], "det%d" % replay, replay, 3)
        cfg.HIP.TRAIN_REPLAY = replay
        seen = _repeat_one_step(net, sess, ts, _blobs(224, 288)[0], 20)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_REPLAY = old + (True,)
    assert len(seen) == 1, len(seen)


def test_fullsize_c5_step_is_deterministic_and_replay_equals_eager(dev):
    """BASELINE configs[4] at full size (ResNet-152, 600 x 1000, 81 classes, A = 12, 256 RoIs; bench.py --config c5's network, weights
    and data layer): 20 repeats of one step from the same state give ONE (loss bits, parameter digest), replayed and eager, and the
    replayed step's result is the eager step's -- the race-freedom of the step's five streams (data-gradient chain, two
    filter-gradient streams, solver, filter preparation) checked by bits on the real launch sizes."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    from model.train_val import synthetic_data_layer
    c = bench.CONFIGS["c5"]
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS, cfg.HIP.TRAIN_REPLAY)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
    try:
        sess = Session(device=dev, seed=cfg.RNG_SEED)
        net = bench.make_net(c)
        net.create_architecture("TRAIN", c["classes"], tag="c5det", anchor_scales=c["scales"], anchor_ratios=bench.ANCHOR_RATIOS)
        sess.init_variables(net.variable_specs())
        ts = TrainState(sess, net, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WEIGHT_DECAY)
        ts.lr = 1e-3
        layer = synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED, image_gain=1 / 256.0)
        blobs = [next(layer) for _ in range(2)]
        cfg.HIP.TRAIN_REPLAY = True
        for i in range(3):                                      # eager, recorded, replayed
            net.train_step(sess, blobs[i % 2], ts)
        assert net.replay_stats == dict(eager=1, recorded=1, replayed=1)
        start = (_save_state(ts), net._sample_seed)
        replayed = _repeat_one_step(net, sess, ts, blobs[1], 20, start)
        assert net.replay_stats["replayed"] == 21
        cfg.HIP.TRAIN_REPLAY = False
        eager = _repeat_one_step(net, sess, ts, blobs[1], 20, start)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS, cfg.HIP.TRAIN_REPLAY = old
    assert len(replayed) == 1 and len(eager) == 1, (len(replayed), len(eager))
    assert replayed == eager


def test_a_new_shape_a_new_learning_rate_or_new_weights_start_a_new_recording(dev):
    from model.config import cfg
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
    try:
        _, _, stats, net, sess, ts = _run(dev, _nets()["res50"], "rpk", True, 4)
        assert stats == dict(eager=1, recorded=1, replayed=2)
        b = _blobs(224, 288)[0]
        ts.lr = 5e-4                                               # cfg.TRAIN.STEPSIZE reached: the rate is a launch argument
        for _ in range(3):
            net.train_step(sess, b, ts)
        assert net.replay_stats == dict(eager=2, recorded=2, replayed=3)
        wide = _blobs(224, 320)[0]                                 # another image shape: other buffers, another list
        for _ in range(3):
            net.train_step(sess, wide, ts)
        assert net.replay_stats == dict(eager=3, recorded=3, replayed=4)
        net.train_step(sess, b, ts)                                # back to the first shape: its recording is still there
        assert net.replay_stats == dict(eager=3, recorded=3, replayed=5)
        sess.load_variables({})                                    # a restore drops everything derived from the variables
        assert not [k for k in sess.graphs if isinstance(k, tuple) and k and k[0] == "train_replay"]
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = old


def test_filter_images_added_after_a_recording_are_kept_current(dev):
    """ADVICE r5 (medium).  A recorded step re-derives the filter images that existed when it was recorded.  A TEST-mode network run on
    the SAME session afterwards derives more (Winograd U of the 3x3 filters, bf16 / fp16 operand planes of layers the training step never
    split): before round 6 the replays kept refreshing the old set only, and the TEST-mode network went on multiplying by images of
    filters the solver had since moved.  Now the recording carries the session's derived-set generation; when it differs the step runs
    eagerly, is recorded again, and EVERY cached image is the image of the filter as it stands -- checked against a fresh derivation."""
    from frcnn_hip import ops
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES, cfg.TEST.RPN_POST_NMS_TOP_N = 64, 0.0, 2, None, 48
    try:
        _, _, stats, net, sess, ts = _run(dev, _nets()["res50"], "gen", True, 4)
        assert stats == dict(eager=1, recorded=1, replayed=2)
        cfg.HIP.TRAIN_REPLAY = True

        def images():
            wino = [k for k in sess.packed if isinstance(k, tuple) and k and k[0] == "wino"]
            return len(sess.h2), len(sess.x3), len(wino)
        before, gen0 = images(), sess.derived_generation()
        tnet = resnetv1(num_layers=50)
        tnet.create_architecture("TEST", 21, tag="gen_test", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        b = _blobs(224, 288)
        tnet.test_image(sess, b[0]["data"], b[0]["im_info"])
        after = images()
        assert sum(after) > sum(before) and sess.derived_generation() != gen0, (before, after)
        for i in range(3):
            net.train_step(sess, b[i % len(b)], ts)
        assert net.replay_stats == dict(eager=1, recorded=2, replayed=4), net.replay_stats      # step 5 recorded afresh, steps 6-7 replayed
        tnet.test_image(sess, b[1]["data"], b[1]["im_info"])           # (orders itself behind the replayed refresh: wait_planes)
        torch.cuda.synchronize()
        n = 0
        for packed, w in sess.h2.values():
            fresh = ops.h2_pack_w(w)
            assert torch.equal(fresh[0], packed[0]) and torch.equal(fresh[1], packed[1])
            n += 1
        for planes, w in sess.x3.values():
            assert torch.equal(ops.gemm_x3_pack(w), planes)
            n += 1
        for key, val in sess.packed.items():
            if isinstance(key, tuple) and key and key[0] == "wino":
                info = sess.conv_info.get(key[1])
                if info is not None and info["w"].dim() == 4 and info["w"].shape[1] == 3:
                    assert torch.equal(ops.winograd_filter_transform_device(info["w"], key[3], False), val[0]), key
                    n += 1
        assert n == sum(images()) and n > sum(before)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES, cfg.TEST.RPN_POST_NMS_TOP_N = old


def test_the_stream_picker_finishes_and_changes_no_bit(dev):
    """cfg.HIP.TRAIN_PICK_STREAMS: the helper slots of the recorded step are re-bound to other physical streams while real steps are timed
    (replay.StreamPicker).  Any binding is correct by construction -- checked by bits against a run that never replays -- and the search
    ends with a binding every later step uses."""
    from model.config import cfg
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_PICK_STREAMS)
    cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_PICK_STREAMS = 64, 0.0, 2
    steps = 2 + 3 * (1 + 4 * 2) + 4
    try:
        l0, d0, _, _, _, _ = _run(dev, _nets()["res50"], "pk0", False, steps)
        l1, d1, s1, net, sess, _ = _run(dev, _nets()["res50"], "pk1", True, steps)
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_PICK_STREAMS = old
    assert s1 == dict(eager=1, recorded=1, replayed=steps - 2)
    assert l1 == l0 and d1 == d0
    assert not sess.picking and len(sess.picked_streams) == 4 and len(sess.pick_log) <= 9 and sess.pick_log[0][0] == "inherited"
