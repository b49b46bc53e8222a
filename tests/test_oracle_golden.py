"""CPU: oracle/frcnn_oracle.py (numpy + C restatement) against the golden vectors that
oracle/gen_golden.py produced by running the REFERENCE's own code (tests/golden/*.npz)."""
import hashlib
import subprocess
import sys
import os

import numpy as np
import pytest

import frcnn_oracle as ora
import synth

f32 = np.float32


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("tag,H,W,scales", [("a9_38x63", 38, 63, (8, 16, 32)), ("a12_38x63", 38, 63, (4, 8, 16, 32)),
                                            ("a15_50x84", 50, 84, (2, 4, 8, 16, 32))])
def test_anchors(golden, tag, H, W, scales):
    g = golden["anchors"]
    base = ora.generate_anchors(ratios=(0.5, 1, 2), scales=scales)
    anc, n = ora.generate_anchors_pre(H, W, 16, scales, (0.5, 1, 2))
    assert np.array_equal(base, g[tag + "_base"])
    assert n == g[tag + "_n"] and anc.dtype == f32
    assert np.array_equal(anc[:64], g[tag + "_first"]) and np.array_equal(anc[-64:], g[tag + "_last"])
    assert np.array_equal(sha(anc), g[tag + "_sha"])


def test_anchor_known_answer():
    # the true output of generate_anchors() (the comment table in the reference,
    # generate_anchors.py:14-39, is the MATLAB 1-based variant of it)
    want = np.array([[-84, -40, 99, 55], [-176, -88, 191, 103], [-360, -184, 375, 199], [-56, -56, 71, 71],
                     [-120, -120, 135, 135], [-248, -248, 263, 263], [-36, -80, 51, 95], [-80, -168, 95, 183],
                     [-168, -344, 183, 359]], dtype=np.float64)
    assert np.array_equal(ora.generate_anchors(), want)
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    assert np.array_equal(anc[-1], np.array([824, 248, 1175, 951], dtype=f32))


def test_codec(golden):
    g = golden["codec"]
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    dec = ora.bbox_transform_inv(anc[g["sel"]], g["deltas"])
    assert np.array_equal(dec, g["decoded"])
    assert np.array_equal(ora.clip_boxes(dec.copy(), np.array([600, 1000], dtype=f32)), g["clipped"])
    gt = synth.gt_boxes(4096, 21, seed=5)
    assert np.array_equal(ora.bbox_transform(g["clipped"], gt[:, :4]), g["encoded"])


@pytest.mark.parametrize("tag,H,W,scales,key,post,info", [
    ("test_38x63_a9", 38, 63, (8, 16, 32), "TEST", 300, (600, 1000, 1.6)),
    ("train_38x63_a9", 38, 63, (8, 16, 32), "TRAIN", 2000, (600, 1000, 1.6)),
    ("test_10x14_a9", 10, 14, (8, 16, 32), "TEST", 300, (160, 224, 1.0)),
    ("test_50x84_a15", 50, 84, (2, 4, 8, 16, 32), "TEST", 1000, (800, 1333, 1.6))])
def test_proposal_layer(golden, tag, H, W, scales, key, post, info):
    g = golden["proposal"]
    A = 3 * len(scales)
    prob, dl = synth.rpn_outputs(H, W, A, seed=3)
    anc, _ = ora.generate_anchors_pre(H, W, 16, scales, (0.5, 1, 2))
    blob, sc = ora.proposal_layer(prob, dl, np.array(info, dtype=f32), key, [16], anc, A, post_nms_topN=post)
    assert np.array_equal(blob, g[tag + "_rois"]) and np.array_equal(sc, g[tag + "_scores"])


def test_proposal_top_layer(golden):
    g = golden["proposal"]
    prob, dl = synth.rpn_outputs(38, 63, 9, seed=3)
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    blob, sc = ora.proposal_top_layer(prob, dl, np.array([600, 1000, 1.6], dtype=f32), [16], anc, 9)
    assert np.array_equal(blob, g["top_38x63_a9_rois"]) and np.array_equal(sc, g["top_38x63_a9_scores"])


# ---- USE_E2E_TF graph (the reference's default): goldens = the reference's *_tf bodies run on oracle/tf_numpy_shim.py
@pytest.mark.parametrize("tag,H,W,scales,post,info", [
    ("38x63_a9", 38, 63, (8, 16, 32), 300, (600, 1000, 1.6)),
    ("10x14_a9", 10, 14, (8, 16, 32), 300, (160, 224, 1.0)),
    ("50x84_a15", 50, 84, (2, 4, 8, 16, 32), 1000, (800, 1333, 1.6))])
def test_proposal_layer_tf(golden, tag, H, W, scales, post, info):
    g = golden["proposal_tf"]
    A = 3 * len(scales)
    prob, dl = synth.rpn_outputs(H, W, A, seed=5)
    assert np.array_equal(np.concatenate([sha(prob), sha(dl)]), g["tf_" + tag + "_in_sha"])
    anc, _ = ora.generate_anchors_pre_tf(H, W, 16, scales, (0.5, 1, 2))
    blob, sc = ora.proposal_layer_tf(prob, dl, np.array(info, dtype=f32), anc, A, post_nms_topN=post, nms_thresh=0.7)
    assert np.array_equal(blob, g["tf_" + tag + "_rois"]) and np.array_equal(sc, g["tf_" + tag + "_scores"])


def test_proposal_top_layer_tf_and_truncated_anchors(golden):
    g = golden["proposal_tf"]
    prob, dl = synth.rpn_outputs(38, 63, 9, seed=5)
    anc, _ = ora.generate_anchors_pre_tf(38, 63, 16)
    blob, sc = ora.proposal_top_layer_tf(prob, dl, np.array([600, 1000, 1.6], dtype=f32), anc, 9)
    assert np.array_equal(blob, g["tf_top_38x63_a9_rois"]) and np.array_equal(sc, g["tf_top_38x63_a9_scores"])
    odd, n = ora.generate_anchors_pre_tf(7, 9, 16, (3, 5, 7), (0.5, 1, 2))
    assert n == 7 * 9 * 9 and np.array_equal(odd[:64], g["anchors_odd_7x9_first"]) and np.array_equal(odd[-64:], g["anchors_odd_7x9_last"])
    full = ora.generate_anchors(ratios=(0.5, 1, 2), scales=(3, 5, 7))
    assert np.any(full != np.trunc(full))                       # the case really exercises the int32 truncation
    assert np.array_equal(odd[:9], np.trunc(full).astype(f32))


def _tf_iou_scalar(a, b):
    """Second, scalar statement of ComputeIOU (TensorFlow r1.2 non_max_suppression_op.cc) in f32."""
    ymin_i, xmin_i, ymax_i, xmax_i = min(a[0], a[2]), min(a[1], a[3]), max(a[0], a[2]), max(a[1], a[3])
    ymin_j, xmin_j, ymax_j, xmax_j = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])
    area_i = f32(f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i))
    area_j = f32(f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return f32(0)
    ih = max(f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f32(0))
    iw = max(f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f32(0))
    inter = f32(ih * iw)
    return f32(inter / f32(f32(area_i + area_j) - inter))


def test_tf_nms_two_statements_agree():
    """tf.image.non_max_suppression is third-party (parity unpinned): the vectorised oracle against a literal scalar
    transcription of the r1.2 loop, incl. degenerate boxes, flipped corners and the strict `>` at equality."""
    for seed, k, thr in ((1, 400, 0.7), (2, 400, 0.3), (3, 64, 0.5)):
        d = synth.random_dets(k, seed=seed, cluster=8)
        boxes, scores = d[:, :4].copy(), d[:, 4].copy()
        boxes[5] = boxes[5][[2, 3, 0, 1]]                     # flipped corners
        boxes[7, 2] = boxes[7, 0]                             # zero area
        order = ora.order_desc(scores)
        sel = []
        for i in order:
            if len(sel) >= 50:
                break
            if all(not (_tf_iou_scalar(boxes[i], boxes[j]) > f32(thr)) for j in reversed(sel)):
                sel.append(int(i))
        got = ora.tf_non_max_suppression(boxes, scores, 50, thr)
        assert got.dtype == np.int32 and got.tolist() == sel
    # equality is NOT suppressed (iou > thr): two boxes with IoU exactly 0.5
    b = np.array([[0, 0, 2, 2], [0, 0, 2, 1]], dtype=f32)
    assert ora.tf_non_max_suppression(b, np.array([0.9, 0.8], dtype=f32), 10, 0.5).tolist() == [0, 1]
    assert ora.tf_non_max_suppression(b, np.array([0.9, 0.8], dtype=f32), 10, 0.49).tolist() == [0]
    assert ora.tf_non_max_suppression(np.zeros((0, 4), f32), np.zeros((0,), f32), 10, 0.5).shape == (0,)


@pytest.mark.parametrize("tag,k,thr,cl", [("u3000_t07", 3000, 0.7, 0), ("c3000_t03", 3000, 0.3, 12),
                                          ("c6000_t07", 6000, 0.7, 40), ("c700_t05", 700, 0.5, 5), ("one", 1, 0.3, 0)])
def test_cpu_nms(golden, tag, k, thr, cl):
    g = golden["nms"]
    d = synth.random_dets(k, seed=11, cluster=cl)
    assert np.array_equal(sha(d), g[tag + "_in_sha"])
    assert np.array_equal(np.array(ora.cpu_nms(d, thr), dtype=np.int32), g[tag + "_keep"])


def test_nms_edges():
    assert ora.nms(np.zeros((0, 5), dtype=f32), 0.3) == []           # nms_wrapper.py:18-19
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [100, 100, 120, 120, 0.7]], dtype=f32)
    assert ora.cpu_nms(d, 0.5) == [0, 2]
    # threshold is compared in double: IoU == (float)0.7 < 0.7 is KEPT
    # boxes 10x10 (area 100) and a box giving inter/union exactly 0.7f is hard to build; instead
    # check the equality rule at a threshold that float32 represents exactly
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], dtype=f32)   # IoU = 50/100 = 0.5 exactly
    assert ora.cpu_nms(d, 0.5) == [0]                                   # suppress when ovr >= thresh
    assert ora.cpu_nms(d, 0.5000001) == [0, 1]


@pytest.mark.parametrize("tag,R,C,W,H", [("voc_300x21", 300, 21, 1000.0, 600.0), ("coco_1000x81", 1000, 81, 1333.0, 800.0)])
def test_perclass(golden, tag, R, C, W, H):
    g = golden["perclass"]
    prob, bp, rois = synth.rcnn_outputs(R, C, seed=7, im_w=W, im_h=H)
    im_shape = (int(H / 1.6), int(W / 1.6), 3)
    sc, boxes = ora.im_detect_post(prob, bp, rois, 1.6, im_shape)
    assert np.array_equal(sha(boxes), g[tag + "_boxes_sha"])
    rec = ora.detections_to_records(ora.test_net_post(sc, boxes, C))
    assert np.array_equal(rec, g[tag + "_records"])
    assert rec.shape[0] >= 100


def test_targets(golden):
    g = golden["targets"]
    d = synth.random_dets(600, seed=13)[:, :4].astype(np.float64)
    q = synth.gt_boxes(12, 21, seed=14).astype(np.float64)
    assert np.array_equal(ora.bbox_overlaps(d, q[:, :4]), g["overlaps"])
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    gt = g["gt"]
    im_info = np.array([600, 1000, 1.6], dtype=f32)
    at = ora.anchor_target_layer(np.zeros((1, 38, 63, 18), dtype=f32), gt, im_info, [16], anc, 9, rng=np.random.RandomState(3))
    for a, n in zip(at, ("at_labels", "at_targets", "at_inside", "at_outside")):
        assert np.array_equal(a, g[n]), n
    prob, dl = synth.rpn_outputs(38, 63, 9, seed=3)
    rois, rsc = ora.proposal_layer(prob, dl, im_info, "TRAIN", [16], anc, 9)
    pt = ora.proposal_target_layer(rois, rsc, gt, 21, rng=np.random.RandomState(3))
    for a, n in zip(pt, ("rois", "scores", "labels", "targets", "inside", "outside")):
        assert np.array_equal(a, g["pt_" + n]), n


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/layer_utils"), reason="reference tree only exists in the build container")
def test_pin_against_live_reference():
    """Runs the reference itself (subprocess: its package names collide with the host mirror)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "gen_golden.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MISMATCH" not in r.stdout


def test_get_image_blob_known_answers():
    """oracle.get_image_blob / cv2_resize_linear (cv2 is third party: parity unpinned) -- the published INTER_LINEAR facts:
    scale 1 is the identity, pixel-centre alignment ((dx+0.5)/fx-0.5), clamped borders, dst = round-half-even(src*fx), and
    the reference's scale rule incl. the MAX_SIZE cap (test.py:45-48)."""
    rng = np.random.RandomState(0)
    im = (rng.rand(375, 500, 3) * 255).astype(np.uint8)
    means = np.array([[[102.9801, 115.9465, 122.7717]]])
    blob, s = ora.get_image_blob(im, means)
    assert blob.shape == (1, 600, 800, 3) and s == 1.6 and blob.dtype == f32
    blob, s = ora.get_image_blob(im[:150], means)                          # 150 x 500: capped by MAX_SIZE
    assert s == 2.0 and blob.shape == (1, 300, 1000, 3)
    assert np.array_equal(ora.cv2_resize_linear(im.astype(f32), 1.0, 1.0), im.astype(f32))
    ramp = np.tile(np.arange(50, dtype=f32)[None, :, None], (40, 1, 3))
    r = ora.cv2_resize_linear(ramp, 1.6, 1.6)
    assert r.shape == (64, 80, 3) and np.allclose(r[10, :4, 0], [0, 0.4375, 1.0625, 1.6875]) and r[10, -1, 0] == 49
    down = ora.cv2_resize_linear(ramp, 0.5, 0.5)
    assert down.shape == (20, 25, 3) and np.allclose(down[0, :3, 0], [0.5, 2.5, 4.5])     # 2:1 -> mean of the two centre taps
    assert ora.cv2_resize_linear(np.zeros((5, 5, 3), f32), 0.5, 0.5).shape == (2, 2, 3)    # 2.5 rounds to even
