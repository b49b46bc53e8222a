"""GPU: the drop-in boundary rows added in round 2 through the C ABI and the host mirrors.

* the reference's TWO suppression rules: cpu_nms `(double)ovr >= thresh` (cpu_nms.pyx:65) and the CUDA kernel /
  py_cpu_nms `ovr > thresh` in float32 (nms_kernel.cu:71, py_cpu_nms.py:35) -- goldens are the reference's own outputs;
* pre_nms_topN <= 0 ("all", proposal_layer.py:35) on a full-size 38x63 map and NMS beyond 16 384 boxes;
* batched launches == per-image launches;
* host-oracle sampling of the target layers and the random fill of proposal_top_layer: numpy's global stream seeded like
  the reference run reproduces the reference's arrays."""
import ctypes

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


NMS_CASES = [("u3000_t07", 3000, 0.7, 0), ("c3000_t03", 3000, 0.3, 12), ("c6000_t07", 6000, 0.7, 40), ("c700_t05", 700, 0.5, 5),
             ("one", 1, 0.3, 0)]


@pytest.mark.parametrize("tag,k,thr,cl", NMS_CASES)
def test_gpu_rule_bit_exact_vs_reference_py_cpu_nms(dev, golden, tag, k, thr, cl):
    from frcnn_hip import NMS_RULE_GPU, ops
    d = synth.random_dets(k, seed=11, cluster=cl)
    keep, num = ops.nms(T(d, dev), thr, rule=NMS_RULE_GPU)
    got = keep[:int(num.item())].cpu().numpy()
    assert np.array_equal(got, golden["nms"][tag + "_keep_gpu"])
    assert got.tolist() == ora.gpu_nms(d, thr)


def test_rules_differ_exactly_at_the_threshold(dev, golden):
    """24 pairs with IoU == 0.5 exactly: cpu_nms suppresses the lower box of every pair, the CUDA rule keeps it."""
    import frcnn_hip
    from frcnn_hip import NMS_RULE_CPU, NMS_RULE_GPU, ops
    d = synth.threshold_pairs()
    g = golden["nms"]
    out = {}
    for rule in (NMS_RULE_CPU, NMS_RULE_GPU):
        keep, num = ops.nms(T(d, dev), 0.5, rule=rule)
        out[rule] = keep[:int(num.item())].cpu().numpy()
    assert np.array_equal(out[NMS_RULE_CPU], g["eq_t05_keep"]) and np.array_equal(out[NMS_RULE_GPU], g["eq_t05_keep_gpu"])
    assert out[NMS_RULE_GPU].size == out[NMS_RULE_CPU].size + 24
    # `_nms` (lib/nms/gpu_nms.hpp:1-2) is the CUDA kernel's entry: sorted input, host pointers -> the GPU rule
    order = ora.order_desc(d[:, 4])
    ds = np.ascontiguousarray(d[order])
    keep_h, n_h = np.zeros(ds.shape[0], dtype=np.int32), ctypes.c_int(0)
    frcnn_hip.lib()._nms(keep_h.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n_h), ds.ctypes.data_as(ctypes.c_void_p), ds.shape[0], 5,
                         ctypes.c_float(0.5), 0)
    assert order[keep_h[:n_h.value]].tolist() == g["eq_t05_keep_gpu"].tolist()


def test_nms_mirrors_pick_the_rule_like_the_reference(dev, golden):
    from model.config import cfg
    from model.nms_wrapper import nms
    from nms.cpu_nms import cpu_nms
    from nms.gpu_nms import gpu_nms
    d = synth.threshold_pairs()
    g = golden["nms"]
    assert cpu_nms(d, 0.5) == g["eq_t05_keep"].tolist() and gpu_nms(d, 0.5) == g["eq_t05_keep_gpu"].tolist()
    old = cfg.USE_GPU_NMS
    try:
        cfg.USE_GPU_NMS = True
        assert nms(d, 0.5) == g["eq_t05_keep_gpu"].tolist() and nms(d, 0.5, force_cpu=True) == g["eq_t05_keep"].tolist()
        cfg.USE_GPU_NMS = False
        assert nms(d, 0.5) == g["eq_t05_keep"].tolist()
        assert nms(np.zeros((0, 5), dtype=f32), 0.5) == []
    finally:
        cfg.USE_GPU_NMS = old


@pytest.mark.parametrize("k,thr,cl,seed", [(20000, 0.7, 300, 31), (40000, 0.5, 2500, 32)])
def test_nms_beyond_16384_boxes(dev, k, thr, cl, seed):
    from frcnn_hip import NMS_RULE_GPU, ops
    d = synth.random_dets(k, seed=seed, cluster=cl)
    keep, num = ops.nms(T(d, dev), thr, max_keep=1500)
    assert keep[:int(num.item())].cpu().numpy().tolist() == ora.cpu_nms(d, thr)[:1500]
    keep, num = ops.nms(T(d, dev), thr, max_keep=1500, rule=NMS_RULE_GPU)
    assert keep[:int(num.item())].cpu().numpy().tolist() == ora.gpu_nms(d, thr)[:1500]


@pytest.mark.parametrize("H,W,scales,post", [(38, 63, (8, 16, 32), 300), (50, 84, (2, 4, 8, 16, 32), 1000)])
def test_proposal_layer_all_mode_at_full_size(dev, H, W, scales, post):
    """pre_nms_topN <= 0 = every anchor goes to NMS (proposal_layer.py:35): 21 546 / 63 000 candidates."""
    from frcnn_hip import ops
    A = 3 * len(scales)
    prob, dl = synth.rpn_outputs(H, W, A, seed=17)
    info = np.array([H * 16, W * 16, 1.0], dtype=f32)
    base = ops.generate_anchors(16, (0.5, 1, 2), scales)
    rois, scores, num = ops.proposal_layer(T(prob, dev), T(dl, dev), info[0], info[1], 16, T(base, dev), 0, post, 0.7)
    anc, _ = ora.generate_anchors_pre(H, W, 16, scales, (0.5, 1, 2))
    wr, ws = ora.proposal_layer(prob, dl, info, "TEST", [16], anc, A, pre_nms_topN=0, post_nms_topN=post)
    n = int(num.item())
    assert n == wr.shape[0]
    assert np.array_equal(scores[:n].cpu().numpy(), ws)
    err = float(np.abs(rois[:n].cpu().numpy() - wr).max())
    print("all-mode %dx%d A=%d: %d proposals, max |box - oracle| = %.3g px" % (H, W, A, n, err))
    assert err <= 1e-4 * max(info[0], info[1])


def test_batched_stages_equal_per_image_calls(dev):
    from frcnn_hip import ops
    B, H, W, A, post = 3, 38, 63, 9, 300
    base = T(ops.generate_anchors(16), dev)
    probs, dls = zip(*[synth.rpn_outputs(H, W, A, seed=40 + b) for b in range(B)])
    prob_b, dl_b = T(np.concatenate(probs), dev), T(np.concatenate(dls), dev)
    rois, scores, num = ops.proposal_layer(prob_b, dl_b, 600, 1000, 16, base, 6000, post, 0.7)
    for b in range(B):
        r1, s1, n1 = ops.proposal_layer(T(probs[b], dev), T(dls[b], dev), 600, 1000, 16, base, 6000, post, 0.7)
        sl = slice(b * post, (b + 1) * post)
        assert int(num[b].item()) == int(n1.item())
        assert torch.equal(scores[sl], s1) and torch.equal(rois[sl, 1:], r1[:, 1:])
        n = int(n1.item())
        assert torch.all(rois[sl][:n, 0] == b) and torch.all(r1[:, 0] == 0)          # rois[:,0] = image index
    # crop_and_resize: box_ind = rois[:,0]
    feat = torch.randn(B, H, W, 256, device=dev)
    crops = ops.crop_and_resize(feat, rois, 16.0, 7)
    for b in range(B):
        sl = slice(b * post, (b + 1) * post)
        r0 = rois[sl].clone()
        r0[:, 0] = 0
        assert torch.equal(crops[sl], ops.crop_and_resize(feat[b], r0, 16.0, 7))
        assert np.array_equal(crops[sl].cpu().numpy(), ora.crop_and_resize(feat[b].cpu().numpy(), r0.cpu().numpy(), 16.0, 7))
    # detect_post
    R, C = 300, 21
    parts = [synth.rcnn_outputs(R, C, seed=50 + b) for b in range(B)]
    prob = T(np.concatenate([p[0] for p in parts]), dev)
    bp = T(np.concatenate([p[1] for p in parts]), dev)
    rr = T(np.concatenate([p[2] for p in parts]), dev)
    nr = torch.tensor([300, 120, 0], dtype=torch.int32, device=dev)
    dets, cnt = ops.detect_post(prob, bp, rr, nr, 1.6, 375, 625, batch=B)
    assert dets.shape == (B, 128, 6)
    for b in range(B):
        d1, c1 = ops.detect_post(T(parts[b][0], dev), T(parts[b][1], dev), T(parts[b][2], dev), nr[b:b + 1].clone(), 1.6, 375, 625)
        assert int(cnt[b].item()) == int(c1.item()) and torch.equal(dets[b], d1)
    # strided record views (frcnn_hip.parallel.new_record): per-image slices contiguous, images 776 floats apart
    from frcnn_hip import parallel
    rec, view = parallel.new_record(dev, batch=B)
    cnt2 = torch.zeros((B,), dtype=torch.int32, device=dev)
    ops.detect_post(prob, bp, rr, nr, 1.6, 375, 625, batch=B, out=view, count=cnt2, max_out=128)
    assert torch.equal(view, dets) and torch.equal(cnt2, cnt)


def test_proposal_top_layer_random_fill_follows_numpy_stream(dev, golden):
    """Fewer anchors than RPN_TOP_N -> npr.choice(length, size=rpn_top_n, replace=True) (proposal_top_layer.py:30-33)."""
    from layer_utils.proposal_top_layer import proposal_top_layer
    prob, dl = synth.rpn_outputs(5, 6, 9, seed=3)
    anc, _ = ora.generate_anchors_pre(5, 6, 16)
    np.random.seed(3)
    rois, scores = proposal_top_layer(prob, dl, np.array([80, 96, 1.0], dtype=f32), [16], anc, 9)
    g = golden["proposal"]
    assert rois.shape == (5000, 5) and np.array_equal(scores, g["topfill_5x6_a9_scores"])
    assert np.allclose(rois, g["topfill_5x6_a9_rois"], rtol=0, atol=1e-4 * 96)


def test_target_layer_mirrors_reproduce_the_reference_run(dev, golden):
    """np.random.seed(3) + the mirrors == the reference's own anchor_target_layer / proposal_target_layer outputs
    (tests/golden/targets.npz): labels, weights, sampled rows bit-exact; regression targets to the device logf."""
    from layer_utils.anchor_target_layer import anchor_target_layer
    from layer_utils.proposal_target_layer import proposal_target_layer
    from model.config import cfg
    g = golden["targets"]
    H, W, A = 38, 63, 9
    anc, _ = ora.generate_anchors_pre(H, W, 16)
    gt = g["gt"]
    old = (cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO)
    try:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 256, 0.0               # experiments/cfgs/res101.yml:9,11
        np.random.seed(3)
        lab, tg, iw, ow = anchor_target_layer(np.zeros((1, H, W, 2 * A), dtype=f32), gt, IM_INFO, [16], anc, A)
        assert lab.shape == g["at_labels"].shape and np.array_equal(lab, g["at_labels"])
        assert np.array_equal(iw, g["at_inside"]) and np.array_equal(ow, g["at_outside"])
        assert np.abs(tg - g["at_targets"]).max() <= 2e-6
        np.random.seed(3)                                                      # the reference run re-seeded here (oracle/gen_golden.py)
        rois, sc, labels, btg, biw, bow = proposal_target_layer(g["pt_in_rois"], g["pt_in_scores"], gt, 21)
        assert np.array_equal(rois, g["pt_rois"]) and np.array_equal(sc, g["pt_scores"]) and np.array_equal(labels, g["pt_labels"])
        assert np.array_equal(biw, g["pt_inside"]) and np.array_equal(bow, g["pt_outside"])
        assert np.abs(btg - g["pt_targets"]).max() <= 2e-5                    # (t - mean) / 0.1: ten times the f32 log error
    finally:
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = old


def test_injected_indices_from_the_reference_run(dev, golden):
    """The C entries alone: the recorded draws of the reference run (at_disable / pt_keep_inds) -> the reference outputs."""
    from frcnn_hip import ops
    g = golden["targets"]
    base = T(ops.generate_anchors(16), dev)
    gt = T(g["gt"], dev)
    lab, tg, iw, ow = ops.anchor_target_layer_inject(gt, 600, 1000, 38, 63, base, T(g["at_disable"], dev))
    assert np.array_equal(lab.cpu().numpy(), g["at_labels"]) and np.array_equal(ow.cpu().numpy(), g["at_outside"])
    assert int((lab == 1).sum().item()) + int((lab == 0).sum().item()) == 256
    out = ops.proposal_target_layer_inject(T(g["pt_in_rois"], dev), T(g["pt_in_scores"].reshape(-1), dev), gt, 21,
                                           T(g["pt_keep_inds"], dev), int(g["pt_n_fg"]))
    assert np.array_equal(out[0].cpu().numpy(), g["pt_rois"]) and np.array_equal(out[2].cpu().numpy(), g["pt_labels"])
    assert np.array_equal(out[4].cpu().numpy(), g["pt_inside"])


def test_target_layers_under_the_reference_s_other_modes(dev, golden):
    """TRAIN.RPN_CLOBBER_POSITIVES, RPN_POSITIVE_WEIGHT 0.3, RPN_BBOX_INSIDE_WEIGHTS, USE_GT and BBOX_INSIDE_WEIGHTS as kernel arguments
    (VERDICT r2 missing #3): the C entries with the reference run's recorded draws, and the py_func mirrors with cfg set like the
    reference's, reproduce tests/golden/targets_modes.npz (the reference's own outputs) -- labels, weights, sampled rows bit-exact."""
    from frcnn_hip import ops
    from layer_utils.anchor_target_layer import anchor_target_layer
    from layer_utils.proposal_target_layer import proposal_target_layer
    from model.config import cfg
    g = golden["targets_modes"]
    H, W, A = 38, 63, 9
    base = T(ops.generate_anchors(16), dev)
    gt = T(g["gt"], dev)
    lab, tg, iw, ow = ops.anchor_target_layer_inject(gt, 600, 1000, H, W, base, T(g["at_disable"], dev), neg_overlap=float(g["neg_ov"]),
                                                     opts=list(g["opts_rpn"]))
    assert np.array_equal(lab.cpu().numpy(), g["at_labels"]) and np.array_equal(ow.cpu().numpy(), g["at_outside"])
    assert np.array_equal(iw.cpu().numpy(), g["at_inside"]) and np.abs(tg.cpu().numpy() - g["at_targets"]).max() <= 2e-6
    assert len(np.unique(g["at_outside"])) == 3                    # 0, p / #positives, (1 - p) / #negatives
    out = ops.proposal_target_layer_inject(T(g["pt_in_rois"], dev), T(g["pt_in_scores"].reshape(-1), dev), gt, 21,
                                           T(g["pt_keep_inds"], dev), int(g["pt_n_fg"]), opts=list(g["opts_roi"]))
    got = [o.cpu().numpy() for o in out]
    assert int((g["pt_keep_inds"] >= g["pt_in_rois"].shape[0]).sum()) > 0            # gt boxes were drawn as RoIs
    assert np.array_equal(got[0], g["pt_rois"]) and np.array_equal(got[1], g["pt_scores"]) and np.array_equal(got[2], g["pt_labels"])
    assert np.array_equal(got[4], g["pt_inside"]) and np.array_equal(got[5], g["pt_outside"])
    assert np.abs(got[3] - g["pt_targets"]).max() <= 2e-5
    # the device-sampled entry accepts the same options: gt rows can be drawn, weights follow the counts
    d = ops.proposal_target_layer(T(g["pt_in_rois"], dev), T(g["pt_in_scores"].reshape(-1), dev), gt, 21, batch_size=256, bg_lo=0.0,
                                  seed=7, opts=list(g["opts_roi"]))
    rois_d, cnt = d[0].cpu().numpy(), d[6].cpu().numpy()
    assert cnt[0] + cnt[1] == 256 and np.all(np.unique(d[4].cpu().numpy()) == np.array([0.0, 0.5, 1.0], dtype=f32))
    gtb = g["gt"][:, :4]
    assert any((np.abs(rois_d[:, 1:5] - b).max(axis=1) == 0).any() for b in gtb)      # with IoU 1 every gt box is a fg candidate
    anc, _ = ora.generate_anchors_pre(H, W, 16)
    t = cfg.TRAIN
    old = (t.BATCH_SIZE, t.BG_THRESH_LO, t.RPN_CLOBBER_POSITIVES, t.RPN_POSITIVE_WEIGHT, t.RPN_BBOX_INSIDE_WEIGHTS, t.USE_GT,
           t.BBOX_INSIDE_WEIGHTS, t.RPN_NEGATIVE_OVERLAP)
    try:
        t.BATCH_SIZE, t.BG_THRESH_LO = 256, 0.0
        t.RPN_CLOBBER_POSITIVES, t.RPN_POSITIVE_WEIGHT, t.RPN_BBOX_INSIDE_WEIGHTS = True, 0.3, (1.0, 0.5, 2.0, 1.0)
        t.USE_GT, t.BBOX_INSIDE_WEIGHTS, t.RPN_NEGATIVE_OVERLAP = True, (1.0, 1.0, 0.5, 0.0), 0.35
        np.random.seed(5)
        lab, tg, iw, ow = anchor_target_layer(np.zeros((1, H, W, 2 * A), dtype=f32), g["gt"], IM_INFO, [16], anc, A)
        assert np.array_equal(lab, g["at_labels"]) and np.array_equal(iw, g["at_inside"]) and np.array_equal(ow, g["at_outside"])
        np.random.seed(5)
        rois, sc, labels, btg, biw, bow = proposal_target_layer(g["pt_in_rois"], g["pt_in_scores"], g["gt"], 21)
        assert np.array_equal(rois, g["pt_rois"]) and np.array_equal(sc, g["pt_scores"]) and np.array_equal(labels, g["pt_labels"])
        assert np.array_equal(biw, g["pt_inside"]) and np.array_equal(bow, g["pt_outside"]) and np.abs(btg - g["pt_targets"]).max() <= 2e-5
    finally:
        (t.BATCH_SIZE, t.BG_THRESH_LO, t.RPN_CLOBBER_POSITIVES, t.RPN_POSITIVE_WEIGHT, t.RPN_BBOX_INSIDE_WEIGHTS, t.USE_GT,
         t.BBOX_INSIDE_WEIGHTS, t.RPN_NEGATIVE_OVERLAP) = old


def test_bbox_reg_false_repeats_the_unregressed_boxes(dev):
    """cfg.TEST.BBOX_REG False (test.py:103-105): pred_boxes = np.tile(rois / scale, (1, C)), no clip; the per-class stage then runs NMS on
    those boxes (frcnn_detect_post with bbox_pred = NULL) like the reference loop on np.tile'd boxes."""
    from frcnn_hip import ops
    rng = np.random.RandomState(4)
    R, C, scale = 300, 21, 1.6
    rois = np.zeros((R, 5), dtype=f32)
    xy = rng.rand(R, 2) * np.array([900.0, 500.0])
    wh = 20 + rng.rand(R, 2) * 300
    rois[:, 1:3], rois[:, 3:5] = xy, xy + wh                          # some boxes reach past the image: they are NOT clipped
    prob = rng.dirichlet(np.ones(C) * 0.3, size=R).astype(f32)
    boxes = ops.im_detect_boxes(T(rois, dev), None, scale, 375, 625, num_classes=C).cpu().numpy()
    # rois f32 / im_scales[0] (np.float64, test.py:58,95) is a float64 division; test_net's hstack(...).astype(float32) rounds it once
    want = np.tile((rois[:, 1:5] / np.float64(scale)).astype(f32), (1, C))
    assert np.array_equal(boxes, want)
    dets, cnt = ops.detect_post(T(prob, dev), None, T(rois, dev), None, scale, 375, 625, nms_thresh=0.3, max_per_image=100)
    rec = dets[:int(cnt.item())].cpu().numpy()
    ref = ora.detections_to_records(ora.test_net_post(prob, want, C))
    assert rec.shape == ref.shape and np.array_equal(rec, ref)


def test_bbox_transform_mirrors_vs_reference_golden(dev, golden):
    from model.bbox_transform import bbox_transform, bbox_transform_inv, clip_boxes
    g = golden["codec"]
    anc, _ = ora.generate_anchors_pre(38, 63, 16)
    dec = bbox_transform_inv(anc[g["sel"]], g["deltas"])
    err = float(np.abs(dec - g["decoded"]).max() / max(1.0, float(np.abs(g["decoded"]).max())))
    print("bbox_transform_inv: max relative error vs the reference %.3g" % err)
    assert dec.dtype == np.float32 and err <= 1e-6
    clipped = clip_boxes(g["decoded"].copy(), np.array([600, 1000], dtype=f32))
    assert np.array_equal(clipped, g["clipped"])
    gtb = synth.gt_boxes(4096, 21, seed=5)
    enc = bbox_transform(g["clipped"], gtb[:, :4])
    assert np.abs(enc - g["encoded"]).max() <= 2e-6 * max(1.0, float(np.abs(g["encoded"]).max()))
    assert bbox_transform_inv(np.zeros((0, 4), dtype=f32), np.zeros((0, 84), dtype=f32)).shape == (0, 84)
