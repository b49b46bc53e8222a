"""GPU end-to-end parity of the re-hosted Network (ResNet-v1 Faster R-CNN, TEST mode) against the
oracle: dense part vs the torch-CPU float64 restatement (1e-4, BASELINE.json), detection stages vs
the pinned numpy/C oracle on identical inputs (bit-exact keep sets)."""
import numpy as np
import pytest
import torch

import frcnn_oracle as ora
from dense_ref import DenseRef

pytestmark = pytest.mark.gpu
SCALES, RATIOS = (4, 8, 16), (0.5, 1, 2)


@pytest.fixture(scope="module")
def small_net(dev):
    from frcnn_hip.runtime import Session
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    cfg.TEST.RPN_POST_NMS_TOP_N = 48
    sess = Session(device=dev, seed=3)
    net = resnetv1(num_layers=50)
    net.create_architecture("TEST", 21, tag="default", anchor_scales=SCALES, anchor_ratios=RATIOS)
    sess.init_variables(net.variable_specs())
    # reference initialisers (network.py:239-240).  Near-ties between scores are harmless here: every
    # decision stage is compared against the oracle on IDENTICAL inputs (the HIP tensors).
    rng = np.random.RandomState(5)
    H, W = 150, 200                              # odd intermediate sizes: 75x100 -> 38x50 -> 19x25 -> 10x13
    image = (rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
    im_info = np.array([H, W, 1.0], dtype=np.float32)
    yield sess, net, image, im_info
    cfg.TEST.RPN_POST_NMS_TOP_N = 300
    cfg.TEST.MODE = "nms"


def rel_err(got, want):
    return float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))


def test_test_image_matches_dense_oracle(small_net):
    sess, net, image, im_info = small_net
    cls_score, cls_prob, bbox_pred, rois = net.test_image(sess, image, im_info)
    assert rois.shape[1] == 5 and 0 < rois.shape[0] <= 48 and cls_prob.shape == (rois.shape[0], 21)
    ref = DenseRef(sess.variables, 50, 21, SCALES, RATIOS).test_image(image, im_info, rois=rois, post=48)
    # what plain float32 arithmetic costs on this graph (torch-CPU f32 vs f64): the HIP f32-MFMA path
    # must be in the same class, not merely under the absolute 1e-4 budget
    ref32 = DenseRef(sess.variables, 50, 21, SCALES, RATIOS, dtype=torch.float32).test_image(image, im_info, rois=rois, post=48)
    for name in ("rpn_cls_score", "rpn_bbox_pred", "cls_score", "bbox_pred"):
        got = net._predictions[name].cpu().numpy()[:ref[name].shape[0]]
        assert rel_err(got, ref[name]) <= 4 * rel_err(ref32[name], ref[name]) + 2e-6, name
    head = net._layers["head"].cpu().numpy()
    assert head.shape == ref["head"].shape == (1, 10, 13, 1024)
    assert rel_err(head, ref["head"]) <= 1e-4
    for name in ("rpn_cls_score", "rpn_cls_prob", "rpn_bbox_pred"):
        assert rel_err(net._predictions[name].cpu().numpy(), ref[name]) <= 1e-4, name
    n = rois.shape[0]
    assert np.array_equal(net._layers["pool5"][:n].cpu().numpy(), ora.crop_and_resize(head[0], rois, 16.0, 7))
    assert rel_err(net._layers["fc7"][:n].cpu().numpy(), ref["fc7"]) <= 1e-4
    assert rel_err(cls_score, ref["cls_score"]) <= 1e-4
    assert np.abs(cls_prob - ref["cls_prob"]).max() <= 1e-4
    assert rel_err(bbox_pred, ref["bbox_pred"]) <= 1e-4
    assert abs(cls_prob.sum(axis=1) - 1).max() < 1e-5


def test_extract_head_gives_the_bits_of_the_full_forward(small_net):
    """Network.extract_head (network.py:452-457 of the reference: the head alone) runs under the same batch-invariant launch plan as
    forward_device, so the same image gives the same head features bit for bit -- also when the pipe rules bite (h2 from 2 tiles)."""
    from model.config import cfg
    sess, net, image, im_info = small_net
    old = cfg.HIP.H2_MIN_TILES
    try:
        for min_tiles in (old, 2):
            cfg.HIP.H2_MIN_TILES = min_tiles
            net.test_image(sess, image, im_info)
            want = net._layers["head"].cpu().numpy().copy()
            got = net.extract_head(sess, image)
            assert np.array_equal(got, want), min_tiles
    finally:
        cfg.HIP.H2_MIN_TILES = old


def test_proposals_match_oracle_on_identical_rpn_outputs(small_net):
    sess, net, image, im_info = small_net
    _, _, _, rois = net.test_image(sess, image, im_info)
    prob = net._predictions["rpn_cls_prob"].cpu().numpy()
    dl = net._predictions["rpn_bbox_pred"].cpu().numpy()
    anchors, _ = ora.generate_anchors_pre(prob.shape[1], prob.shape[2], 16, SCALES, RATIOS)
    want, _ = ora.proposal_layer(prob, dl, im_info, "TEST", [16], anchors, 9, post_nms_topN=48)
    assert want.shape == rois.shape and np.allclose(rois, want, rtol=0, atol=1e-3)


def test_graph_replay_is_deterministic_and_device_post_matches(small_net):
    sess, net, image, im_info = small_net
    a = net.test_image(sess, image, im_info)
    b = net.test_image(sess, image, im_info)             # second call = pure hipGraph replay
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    img_d = net._stage_image(sess, image)
    dets, cnt = net.detect_device(sess, img_d, im_info, (150, 200))
    n = int(cnt.item())
    sc, boxes = ora.im_detect_post(a[1], a[2], a[3], 1.0, (150, 200, 3))
    want = ora.detections_to_records(ora.test_net_post(sc, boxes, 21))
    got = dets[:n].cpu().numpy()
    assert n == want.shape[0] > 0
    assert np.array_equal(got[:, 4:], want[:, 4:]) and np.allclose(got[:, :4], want[:, :4], rtol=0, atol=1e-3)


def test_eager_equals_graph_and_top_mode(small_net):
    from model.config import cfg
    sess, net, image, im_info = small_net
    g = net.test_image(sess, image, im_info)
    img_d = net._stage_image(sess, image)
    p = net.forward_device(sess, img_d, im_info, use_graph=False)
    n = int(net._num_rois.item())
    assert np.array_equal(p["cls_prob"][:n].cpu().numpy(), g[1])
    cfg.TEST.MODE, cfg.TEST.RPN_TOP_N = "top", 200
    try:
        _, cls_prob, _, rois = net.test_image(sess, image, im_info)
        assert rois.shape == (200, 5) and cls_prob.shape == (200, 21)
        prob = net._predictions["rpn_cls_prob"].cpu().numpy()
        dl = net._predictions["rpn_bbox_pred"].cpu().numpy()
        anchors, _ = ora.generate_anchors_pre(prob.shape[1], prob.shape[2], 16, SCALES, RATIOS)
        want, _ = ora.proposal_top_layer(prob, dl, im_info, [16], anchors, 9, rpn_top_n=200)
        assert np.allclose(rois, want, rtol=0, atol=1e-3)
    finally:
        cfg.TEST.MODE, cfg.TEST.RPN_TOP_N = "nms", 5000


@pytest.mark.parametrize("arch", ["vgg16", "mobile"])
def test_vgg16_and_mobilenet_match_dense_oracle(dev, arch):
    """SURVEY.md 8a rows 2/10 (VGG16, MobileNet-v1 backbones + tails): same parity bar as ResNet."""
    from dense_ref import VGG16Ref, MobileNetRef
    from frcnn_hip.runtime import Session
    from model.config import cfg
    from nets.mobilenet_v1 import mobilenetv1
    from nets.vgg16 import vgg16
    prev_post = cfg.TEST.RPN_POST_NMS_TOP_N
    cfg.TEST.RPN_POST_NMS_TOP_N = 24
    try:
        sess = Session(device=dev, seed=7)
        net = vgg16() if arch == "vgg16" else mobilenetv1()
        net.create_architecture("TEST", 21, tag=arch, anchor_scales=SCALES, anchor_ratios=RATIOS)
        sess.init_variables(net.variable_specs())
        rng = np.random.RandomState(11)
        H, W = 121, 170                                      # odd sizes exercise the SAME / conv2d_same rules
        image = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 64.0 if arch == "vgg16" else 1.0)
        im_info = np.array([H, W, 1.0], dtype=np.float32)
        cls_score, cls_prob, bbox_pred, rois = net.test_image(sess, image, im_info)
        Ref = VGG16Ref if arch == "vgg16" else MobileNetRef
        ref = Ref(sess.variables, 21, SCALES, RATIOS).test_image(image, im_info, rois=rois, post=24)
        ref32 = Ref(sess.variables, 21, SCALES, RATIOS, dtype=torch.float32).test_image(image, im_info, rois=rois, post=24)
        head = net._layers["head"].cpu().numpy()
        assert head.shape == ref["head"].shape
        assert rel_err(head, ref["head"]) <= 1e-4
        n = rois.shape[0]
        assert n > 0
        assert np.array_equal(net._layers["pool5"][:n].cpu().numpy(), ora.crop_and_resize(head[0], rois, 16.0, 7, max_pool=True))
        for name, got in (("rpn_cls_prob", net._predictions["rpn_cls_prob"].cpu().numpy()), ("rpn_bbox_pred", net._predictions["rpn_bbox_pred"].cpu().numpy()),
                          ("cls_score", cls_score), ("bbox_pred", bbox_pred)):
            assert rel_err(got, ref[name][:got.shape[0]]) <= max(1e-4, 4 * rel_err(ref32[name][:got.shape[0]], ref[name][:got.shape[0]])), name
        assert np.abs(cls_prob - ref["cls_prob"]).max() <= 1e-4
    finally:
        cfg.TEST.RPN_POST_NMS_TOP_N = prev_post


def test_model_test_module_matches_reference_loop(small_net):
    """model.test.im_detect / detect (lib/model/test.py:86-107,156-180) vs the oracle on identical inputs."""
    from model.test import detect, im_detect
    sess, net, image, im_info = small_net
    scores, pred_boxes = im_detect(sess, net, image, 1.0, (150, 200, 3))
    _, cls_prob, bbox_pred, rois = net.test_image(sess, image, im_info)
    want_s, want_b = ora.im_detect_post(cls_prob, bbox_pred, rois, 1.0, (150, 200, 3))
    assert np.array_equal(scores, want_s) and np.allclose(pred_boxes, want_b, rtol=0, atol=1e-3)
    per_class = detect(sess, net, image, 1.0, (150, 200))
    want = ora.test_net_post(want_s, want_b, 21)
    for j in range(1, 21):
        assert per_class[j].shape == want[j].shape
        assert np.array_equal(per_class[j][:, 4], want[j][:, 4]) and np.allclose(per_class[j][:, :4], want[j][:, :4], atol=1e-3)


def test_fused_tail_entry_equals_reference_order(small_net):
    """resnetv1._fused_tail_entry: crop commuted past block4/unit_1's 1x1 convs == the reference op order
    (up to f32 rounding) and within the same 1e-4 budget against the float64 oracle."""
    sess, net, image, im_info = small_net
    base = net.test_image(sess, image, im_info)
    net._fuse_tail_entry = True
    try:
        fused = net.test_image(sess, image, im_info)
    finally:
        net._fuse_tail_entry = False
    assert np.array_equal(base[3], fused[3])                                     # same rois
    for a, b in zip(base[:3], fused[:3]):
        assert rel_err(b, a) <= 2e-5
    ref = DenseRef(sess.variables, 50, 21, SCALES, RATIOS).test_image(image, im_info, rois=fused[3], post=48)
    assert rel_err(fused[0], ref["cls_score"]) <= 1e-4 and np.abs(fused[1] - ref["cls_prob"]).max() <= 1e-4
    assert rel_err(fused[2], ref["bbox_pred"]) <= 1e-4


def test_batched_forward_equals_single_image_forward(small_net):
    """A batch is B independent images whose dense layers share launches; the reference is strictly batch-1 (lib/model/test.py:88).
    The same image must give the same BITS alone and in any slot of a batch -- under the shipped configuration (h2 + x3 + Winograd),
    with h2 forced onto every eligible layer (H2_MIN_TILES = 1), with x3 only, and with every product on the f32 MFMA: the pipe a GEMM
    runs on and the split-K plan of a convolution follow the per-image shape (network.py _plan_rows, conv_igemm.hip plan_splits), and
    the tile configurations of one kernel all multiply in the same order."""
    sess, net, image, im_info = small_net
    rng = np.random.RandomState(9)
    from model.config import cfg
    img2 = (rng.rand(1, 150, 200, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
    batch = net._stage_image(sess, np.concatenate([image, img2, image], axis=0))
    keep = (cfg.HIP.MFMA_X3, cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES)
    keys = ("rois", "cls_score", "bbox_pred", "rpn_cls_prob", "rpn_bbox_pred", "cls_prob")
    try:
        for x3, h2, mint in ((keep[0], keep[1], keep[2]), (True, True, 1), (True, False, keep[2]), (False, False, keep[2])):
            cfg.HIP.MFMA_X3, cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES = x3, h2, mint
            singles = []
            for im in (image, img2):
                p = net.forward_device(sess, net._stage_image(sess, im), im_info)
                torch.cuda.synchronize()
                singles.append({k: p[k].cpu().numpy().copy() for k in keys})
            p = net.forward_device(sess, batch, im_info)
            torch.cuda.synchronize()
            per = net._rois_per_image
            assert p["rois"].shape[0] == 3 * per and net._num_rois.shape[0] == 3
            what = "x3 %s h2 %s min tiles %d" % (x3, h2, mint)
            for b, want in enumerate((singles[0], singles[1], singles[0])):
                for k in ("rpn_cls_prob", "rpn_bbox_pred"):
                    assert np.array_equal(p[k][b:b + 1].cpu().numpy(), want[k]), (what, k, b)
                sl = slice(b * per, (b + 1) * per)
                got_rois = p["rois"][sl].cpu().numpy()
                nb = int(net._num_rois[b].item())
                assert np.all(got_rois[:nb, 0] == b) and np.all(want["rois"][:, 0] == 0)       # rois[:,0] = image index of the batch
                assert np.array_equal(got_rois[:, 1:], want["rois"][:, 1:]), (what, "rois", b)
                for k in ("cls_score", "bbox_pred", "cls_prob"):
                    assert np.array_equal(p[k][sl].cpu().numpy(), want[k]), (what, k, b)
            d, c = net.detect_device(sess, batch, im_info, (150, 200))
            c = c.cpu().numpy()
            assert d.shape[0] == 3 and np.all(c > 0) and c[0] == c[2]
            assert np.array_equal(d[0, :c[0]].cpu().numpy(), d[2, :c[2]].cpu().numpy())        # same image, same batch -> identical
            d1, c1 = net.detect_device(sess, net._stage_image(sess, image), im_info, (150, 200))
            assert int(c1.cpu().numpy()[0]) == c[0] and np.array_equal(d1[:c[0]].cpu().numpy(), d[0, :c[0]].cpu().numpy()), what   # one image: dets [max_out, 6]
    finally:
        cfg.HIP.MFMA_X3, cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES = keep


def test_h2_path_end_to_end_meets_the_f32_bounds(dev):
    """cfg.HIP.MFMA_H2 forced onto a small network (H2_MIN_TILES = 1: every plain GEMM with Cin, Cout % 128 == 0 runs in
    frcnn_gemm_h2, the Winograd transforms and GEMM epilogues emit the operand planes, un-planed inputs are split lazily; 128 x 192
    image so that every layer's row count is a multiple of 4): the same bounds against the float64 oracle as the f32-MFMA path,
    h2 launches really happened, and tensors handed over as planes only are never read as float32."""
    from frcnn_hip.runtime import Session
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    keep = (cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N = True, 1, True, 48
    try:
        sess = Session(device=dev, seed=11)
        net = resnetv1(num_layers=50)
        net.create_architecture("TEST", 21, tag="h2small", anchor_scales=SCALES, anchor_ratios=RATIOS)
        sess.init_variables(net.variable_specs())
        rng = np.random.RandomState(6)
        H, W = 128, 192
        image = (rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
        im_info = np.array([H, W, 1.0], dtype=np.float32)
        for fuse in (False, True):
            net._fuse_tail_entry = fuse
            cls_score, cls_prob, bbox_pred, rois = net.test_image(sess, image, im_info)
            head = net._layers["head"].cpu().numpy()
            rpn = {k: net._predictions[k].cpu().numpy() for k in ("rpn_cls_score", "rpn_bbox_pred")}
            sess.profile = []
            net.forward_device(sess, net._stage_image(sess, image), im_info)
            torch.cuda.synchronize()
            tags = [t[0] for t in sess.profile]
            sess.profile = None
            n_h2, n_conv = sum(1 for t in tags if t.startswith("conv:h2:")), sum(1 for t in tags if t.startswith("conv:"))
            print("fused tail entry %s: %d h2 launches of %d conv launches, %d lazy splits" % (fuse, n_h2, n_conv, tags.count("op:h2_split")))
            assert n_h2 >= 30                      # block2-4 bottlenecks + RPN 3x3 (block1 and the heads are not eligible)
            ref = DenseRef(sess.variables, 50, 21, SCALES, RATIOS).test_image(image, im_info, rois=rois, post=48)
            ref32 = DenseRef(sess.variables, 50, 21, SCALES, RATIOS, dtype=torch.float32).test_image(image, im_info, rois=rois, post=48)
            assert rel_err(head, ref["head"]) <= 1e-4
            for name, got in (("rpn_cls_score", rpn["rpn_cls_score"]), ("rpn_bbox_pred", rpn["rpn_bbox_pred"]), ("cls_score", cls_score),
                              ("bbox_pred", bbox_pred)):
                e, e32 = rel_err(got, ref[name]), rel_err(ref32[name], ref[name])
                print("  %-14s |h2 path - f64| = %.2e   (torch f32 control %.2e)" % (name, e, e32))
                assert e <= 4 * e32 + 2e-6, name
            assert np.abs(cls_prob - ref["cls_prob"]).max() <= 1e-4
        sess.close()
    finally:
        cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N = keep


def test_direct_conv_path_equals_winograd_path(small_net):
    """cfg.HIP.WINOGRAD only changes HOW the 3x3 stride-1 convolutions are evaluated (F(2x2,3x3), exact algebra):
    both settings meet the float64 oracle bound and agree with each other to f32 rounding."""
    from model.config import cfg
    sess, net, image, im_info = small_net
    assert cfg.HIP.WINOGRAD
    wino = net.test_image(sess, image, im_info)
    wino_rpn = {k: net._predictions[k].cpu().numpy().copy() for k in ("rpn_cls_score", "rpn_bbox_pred")}
    n_wino = sum(1 for k in sess.packed if isinstance(k, tuple) and k[0] == "wino" and k[-1] == cfg.HIP.WINOGRAD_M)
    assert n_wino > 0                                                             # the path really ran
    cfg.HIP.WINOGRAD = False
    try:
        direct = net.test_image(sess, image, im_info)
        direct_rpn = {k: net._predictions[k].cpu().numpy().copy() for k in wino_rpn}
    finally:
        cfg.HIP.WINOGRAD = True
    for k in wino_rpn:
        assert rel_err(wino_rpn[k], direct_rpn[k]) <= 2e-5, k
    # this fixture's RPN deltas are large (exp() amplifies f32 rounding of the deltas into ~1e-2 px on the boxes)
    assert np.allclose(direct[3], wino[3], rtol=0, atol=5e-2)
    ref = DenseRef(sess.variables, 50, 21, SCALES, RATIOS).test_image(image, im_info, rois=direct[3], post=48)
    assert rel_err(direct[0], ref["cls_score"]) <= 1e-4 and rel_err(direct[2], ref["bbox_pred"]) <= 1e-4
    if np.array_equal(direct[3], wino[3]):
        for a, b in zip(direct[:3], wino[:3]):
            assert rel_err(b, a) <= 2e-5


def test_use_e2e_tf_graph_semantics(small_net):
    """cfg.USE_E2E_TF (the reference's default graph): truncated anchors + proposal_layer_tf.  The proposals of the device
    chain equal the oracle's proposal_layer_tf on the device's own RPN outputs; the rest of the chain is unchanged."""
    from model.config import cfg
    sess, net, image, im_info = small_net
    cfg.USE_E2E_TF = True
    try:
        cls_score, cls_prob, bbox_pred, rois = net.test_image(sess, image, im_info)
        prob = net._predictions["rpn_cls_prob"].cpu().numpy()
        dl = net._predictions["rpn_bbox_pred"].cpu().numpy()
        n = int(net._num_rois[0].item())
    finally:
        cfg.USE_E2E_TF = False
    H, W = prob.shape[1:3]
    anc, _ = ora.generate_anchors_pre_tf(H, W, 16, SCALES, RATIOS)
    want, _ = ora.proposal_layer_tf(prob, dl, im_info, anc, len(SCALES) * len(RATIOS), post_nms_topN=48, nms_thresh=0.7)
    assert n == want.shape[0] and np.allclose(rois[:n], want, rtol=0, atol=2e-2)
    ref = DenseRef(sess.variables, 50, 21, SCALES, RATIOS).test_image(image, im_info, rois=rois, post=48)
    assert rel_err(cls_score, ref["cls_score"]) <= 1e-4 and rel_err(bbox_pred, ref["bbox_pred"]) <= 1e-4


def test_raw_bgr_image_entry_points(small_net):
    """model.test.im_detect_bgr / detect_bgr (the reference's `im_detect(sess, net, im)` signature): uint8 BGR in, the
    device _get_image_blob (frcnn_prep_image) in front of the same chain == oracle blob fed through the blob entry points."""
    from model.config import cfg
    from model.test import detect, detect_bgr, im_detect, im_detect_bgr
    sess, net, _, _ = small_net
    rng = np.random.RandomState(11)
    im = (rng.rand(120, 160, 3) * 255).astype(np.uint8)
    old = (cfg.TEST.SCALES, cfg.TEST.MAX_SIZE)
    cfg.TEST.SCALES, cfg.TEST.MAX_SIZE = (150,), 220
    try:
        blob, im_scale = ora.get_image_blob(im, cfg.PIXEL_MEANS, 150, 220)
        assert blob.shape == (1, 150, 200, 3) and im_scale == 1.25
        s1, b1 = im_detect_bgr(sess, net, im)
        s2, b2 = im_detect(sess, net, blob, im_scale, im.shape)
        assert np.array_equal(s1, s2) and np.array_equal(b1, b2)
        d1 = detect_bgr(sess, net, torch.from_numpy(im).to(sess.device))          # device-resident source image
        d2 = detect(sess, net, blob, im_scale, im.shape[:2])
        for j in range(1, 21):
            assert np.array_equal(d1[j], d2[j])
    finally:
        cfg.TEST.SCALES, cfg.TEST.MAX_SIZE = old


def test_tools_test_net_on_a_voc_devkit_and_checkpoint(dev, tmp_path, capsys):
    """tools/test_net.py --imdb voc_2007_test --model <TF V2 checkpoint>: JPEGs -> raw-image device path -> results files
    -> VOC07 AP, with the weights restored from a checkpoint written without TensorFlow."""
    import importlib.util
    import os
    import pickle
    import sys
    from PIL import Image
    import gen_golden_eval as gge
    from frcnn_hip.runtime import VariableStore
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gt, dets = gge.synth_arrays(3, 4)
    data_dir = tmp_path / "data"
    voc = data_dir / "VOCdevkit2007" / "VOC2007"
    index, _, _, _ = gge.build_devkit(str(voc), gt, dets, 4)
    os.makedirs(str(voc / "JPEGImages"))
    rng = np.random.RandomState(0)
    for name in index:
        Image.fromarray((rng.rand(120, 160, 3) * 255).astype(np.uint8)).save(str(voc / "JPEGImages" / (name + ".jpg")))
    net = resnetv1(num_layers=50)
    net.create_architecture("TEST", 21, tag="default", anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    store = VariableStore(seed=11)
    store.init_variables(net.variable_specs())
    ckpt = store.save(str(tmp_path / "res50_faster_rcnn_iter_1.ckpt"), {"global_step": np.array(1, dtype=np.int64)})
    tools = os.path.join(root, "tf-faster-rcnn_amd", "tools")
    sys.path.insert(0, tools)
    spec = importlib.util.spec_from_file_location("frcnn_tools_test_net", os.path.join(tools, "test_net.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = (cfg.DATA_DIR, cfg.ROOT_DIR)
    try:
        rc = mod.main(["--imdb", "voc_2007_test", "--net", "res50", "--model", ckpt, "--comp", "--set", "DATA_DIR", str(data_dir),
                       "ROOT_DIR", str(tmp_path)])
    finally:
        cfg.DATA_DIR, cfg.ROOT_DIR = old
    assert rc == 0
    out = capsys.readouterr().out
    assert "Loaded." in out and "im_detect: 4/4" in out and "Mean AP = " in out
    out_dir = tmp_path / "output" / "res50" / "voc_2007_test" / "default"
    boxes = pickle.load(open(str(out_dir / "detections.pkl"), "rb"))
    assert len(boxes) == 21 and len(boxes[1]) == 4 and sum(len(boxes[j][0]) for j in range(1, 21)) > 0
    assert os.path.isfile(str(data_dir / "VOCdevkit2007" / "results" / "VOC2007" / "Main" / "comp4_det_test_aeroplane.txt"))
    assert os.path.isfile(str(out_dir / "aeroplane_pr.pkl"))
    # tools/reval.py on that run: --nms re-applies the (device) NMS; the lists were NMS'd at the same threshold already
    spec = importlib.util.spec_from_file_location("frcnn_tools_reval", os.path.join(tools, "reval.py"))
    reval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(reval)
    from model.test import apply_nms
    again = apply_nms(boxes, cfg.TEST.NMS)
    for j in range(1, 21):
        for i in range(4):
            a, b = np.asarray(boxes[j][i], dtype=np.float32).reshape(-1, 5), again[j][i]
            valid = a[(a[:, 2] > a[:, 0]) & (a[:, 3] > a[:, 1])]                 # clipped boxes can be degenerate (test.py:124)
            assert (len(valid) == 0 and len(b) == 0) or np.array_equal(valid, b)
    old = (cfg.DATA_DIR, cfg.ROOT_DIR)
    try:
        assert reval.main([str(out_dir), "--imdb", "voc_2007_test", "--nms", "--comp", "--set", "DATA_DIR", str(data_dir)]) == 0
    finally:
        cfg.DATA_DIR, cfg.ROOT_DIR = old
    assert "Applying NMS to all detections" in capsys.readouterr().out


def test_fused_tail_mean_matches_the_unfused_path(small_net):
    """cfg.HIP.FUSE_TAIL_MEAN (default: the last conv3 + reduce_mean in one frcnn_gemm_h2_mean launch) against the unfused path (conv3
    writes its tensor, frcnn_spatial_mean reads it): the same tensors up to f32 summation order."""
    from model.config import cfg
    sess, net, image, im_info = small_net
    assert cfg.HIP.FUSE_TAIL_MEAN
    fused = [a.copy() for a in net.test_image(sess, image, im_info)]
    try:
        cfg.HIP.FUSE_TAIL_MEAN = False
        base = net.test_image(sess, image, im_info)
    finally:
        cfg.HIP.FUSE_TAIL_MEAN = True
    assert np.array_equal(fused[3], base[3])                                     # same proposals
    for a, b in ((fused[0], base[0]), (fused[2], base[2])):                      # cls_score, bbox_pred (this fixture's logits are O(1e3))
        assert rel_err(a, b) <= 2e-5
    # the tail's conv2 on the direct kernel and no lazy split: no operand planes reach the last conv3 -> the graph falls back to the
    # unfused pair instead of refusing to build (h2 from 2 tiles so that the fused form WOULD apply at this toy size)
    keep = (cfg.HIP.WINOGRAD_DIRECT_SCOPES, cfg.HIP.H2_LAZY_SPLIT, cfg.HIP.H2_MIN_TILES)
    try:
        cfg.HIP.WINOGRAD_DIRECT_SCOPES, cfg.HIP.H2_LAZY_SPLIT, cfg.HIP.H2_MIN_TILES = ("block4",), False, 2
        direct = net.test_image(sess, image, im_info)
    finally:
        cfg.HIP.WINOGRAD_DIRECT_SCOPES, cfg.HIP.H2_LAZY_SPLIT, cfg.HIP.H2_MIN_TILES = keep
    for a, b in ((direct[0], base[0]), (direct[2], base[2])):   # (block4's 3x3 layers direct instead of Winograd, conv3 on another pipe: f32 rounding
        assert rel_err(a, b) <= 2e-4                             # of a different operation order, not the fused form's 2e-5)


def test_h2_static_filter_criterion_falls_back_to_x3(dev):
    """A consumer filter whose entries for one input channel are below 2^-18 of the rest (the signature of an activation channel
    2^18 times the others) keeps that layer, and the producer feeding it, off the block-scaled fp16x2 format: the exact x3 split runs
    instead (Session.h2_channel_spread, Network._h2_eligible); 2^-17 still runs in frcnn_gemm_h2.  Both give the f32-class result."""
    from frcnn_hip.runtime import Session
    from model.config import cfg
    from nets.resnet_v1 import resnetv1
    keep = (cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N = True, 1, True, 48
    try:
        rng = np.random.RandomState(5)
        image = (rng.rand(1, 128, 192, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
        im_info = np.array([128, 192, 1.0], dtype=np.float32)
        tags = {}
        for k in (17, 20):
            sess = Session(device=dev, seed=11)
            net = resnetv1(num_layers=50)
            net.create_architecture("TEST", 21, tag="h2crit%d" % k, anchor_scales=SCALES, anchor_ratios=RATIOS)
            sess.init_variables(net.variable_specs())
            u = "resnet_v1_50/block3/unit_2/bottleneck_v1"
            s_ = np.float32(2.0 ** k)                                    # channel 5 of conv1's output is 2^k times its base value ...
            sess.variables[u + "/conv1/BatchNorm/gamma"][5] *= s_
            sess.variables[u + "/conv1/BatchNorm/beta"][5] *= s_
            sess.variables[u + "/conv2/weights"][:, :, 5, :] /= s_       # ... and conv2's filters undo it: same function
            sess._drop_derived()
            sess.profile = []
            net.forward_device(sess, net._stage_image(sess, image), im_info, use_graph=False)
            torch.cuda.synchronize()
            tags[k] = [t[0] for t in sess.profile]
            sess.profile = None
            spread = sess.h2_channel_spread(u + "/conv2")
            assert (spread < 2.0 ** -18) == (k == 20), (k, spread)
        assert ("conv:h2:" + u + "/conv2") in tags[17] and ("conv:h2:" + u + "/conv2") not in tags[20]
        assert any(t.endswith(u + "/conv2") and not t.startswith("conv:h2:") for t in tags[20])
        assert ("conv:h2:" + "resnet_v1_50/block3/unit_3/bottleneck_v1/conv2") in tags[20]          # the other layers are untouched
    finally:
        cfg.HIP.MFMA_H2, cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_LAZY_SPLIT, cfg.TEST.RPN_POST_NMS_TOP_N = keep
