#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/full_c5_train.npz: the float64 reference of ONE full-size training step's
losses and gradients for BASELINE.json configs[4] (ResNet-152, 600x1000, A = 12, 81 classes, 256 RoIs).

    python oracle/gen_fullsize_train.py            # a minute or two on host cores, ~10 GB of RAM

Reference semantics: /root/reference/lib/nets/network.py:264-321 (the four losses), lib/model/train_val.py:116-153 (gradients of
their sum w.r.t. the trainable variables; RESNET.FIXED_BLOCKS = 1: conv1 + block1 frozen, batch norm frozen everywhere,
resnet_v1.py:88-113), res101.yml / res152 experiments: TRAIN.BATCH_SIZE 256, BG_THRESH_LO 0.0.  Dense arithmetic:
oracle/dense_ref.py::TrainRef in torch float64 autograd (PARITY UNPINNED for the TF/slim semantics, see that file).  The sampled
quantities -- proposals, anchor targets, proposal targets -- come from the PINNED numpy/C oracle (proposal_layer /
anchor_target_layer / proposal_target_layer with a seeded RandomState) on the float32 casts of the float64 RPN tensors, and are
stored in the fixture: the GPU test feeds exactly these constants to the device graph (they carry no gradient in the reference
either: py_func outputs, network.py:153), so losses and gradients are compared on identical samples.

Weights: bench.py's He filters and BN gammas ("damped": the last BN of every residual branch has gamma ~ U(0.1, 0.3)) with the
frozen statistics CALIBRATED to the actual per-channel statistics on this image (block4's on the 256 sampled RoIs), so every
activation is O(1) like in a trained network; RPN and class heads rescaled to the statistics SURVEY.md 8d prescribes.
(`--gammas calibrated`, every gamma ~ U(0.5, 1.5), makes the random 152-layer graph chaotic in the BACKWARD direction: torch
float32 autograd is then 4e-2 ... 8e-2 of |g|max away from float64 on the block2 / block3 filters -- nothing to compare against.
With the damped gammas the float32 control is 1e-6 on the heads and 2e-4 ... 8e-3 on the deep filters; both numbers are stored.)
Stored gradients are sub-sampled with a fixed stride (layout [Cout][KH][KW][Cin], the device's master-filter layout); a float32
control (the same graph in torch float32 autograd) is stored beside them so the test can print |device - control|.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import frcnn_oracle as ora  # noqa: E402
import fullsize as fs  # noqa: E402
import synth  # noqa: E402
from dense_ref import DenseRef, TrainRef  # noqa: E402
from gen_fullsize import calib_class  # noqa: E402


def grad_sample(g_hwio_or_mat, stride):
    """torch gradient of a TF-layout variable (HWIO filter or [in,out] matrix) -> the device's master layout [Cout][KH][KW][Cin]
    flattened, every `stride`-th element."""
    g = np.asarray(g_hwio_or_mat)
    if g.ndim == 2:
        g = g[None, None]
    if g.ndim == 4:
        g = np.transpose(g, (3, 0, 1, 2))
    flat = np.ascontiguousarray(g).ravel()
    return flat[::stride], float(np.abs(flat).max())


def collect(ref, scope, scopes):
    out = {}
    for key, (sc, stride) in scopes.items():
        name = scope + sc
        g = ref._cache[name].grad
        assert g is not None, name
        s, amax = grad_sample(g.numpy(), stride)
        out[key] = (s, amax)
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--gammas", choices=["damped", "calibrated"], default=fs.TRAIN_CONFIGS["c5"]["gammas"])
    args = ap.parse_args()
    c = fs.TRAIN_CONFIGS["c5"]
    t0 = time.time()
    net, v = fs.base_variables("c5", args.gammas)
    scope = net._scope
    image = fs.synth_image(c)
    im_info = np.array([c["H"], c["W"], c["scale"]], dtype=np.float32)
    gt = synth.gt_boxes(c["num_gt"], c["classes"], seed=3, im_w=float(c["W"]), im_h=float(c["H"]))
    A = len(c["scales"]) * len(c["ratios"])
    # ---- pass 1 (no autograd): calibrate the frozen BN statistics, scale the heads, draw the samples
    ref = calib_class(DenseRef)(v, c["layers"], c["classes"], c["scales"], c["ratios"], dtype=torch.float64)
    ref.bn_order = []
    ref.calibrate = True
    fx = {}
    with torch.no_grad():
        feat = ref.head(image)
        score, prob, bbox = ref.rpn(feat)
        s_cls = np.float32(1.0 / max(float(np.std(score)), 1e-12))
        s_box = np.float32(0.2 / max(float(np.std(bbox)), 1e-12))
        fx.update(rpn_cls_scale=s_cls, rpn_box_scale=s_box)
        v[scope + "/rpn_cls_score/weights"] = (v[scope + "/rpn_cls_score/weights"] * s_cls).astype(np.float32)
        v[scope + "/rpn_bbox_pred/weights"] = (v[scope + "/rpn_bbox_pred/weights"] * s_box).astype(np.float32)
        ref._cache.pop(scope + "/rpn_cls_score/weights", None)
        ref._cache.pop(scope + "/rpn_bbox_pred/weights", None)
        score, prob, bbox = ref.rpn(feat)
        H, W = feat.shape[2], feat.shape[3]
        anchors, _ = ora.generate_anchors_pre(H, W, 16, c["scales"], c["ratios"])
        rois_all, sc_all = ora.proposal_layer(prob.astype(np.float32), bbox.astype(np.float32), im_info, "TRAIN", [16], anchors, A)
        rng = np.random.RandomState(3)                      # cfg.RNG_SEED; call order as in the graph: anchor targets first (network.py:342)
        at = ora.anchor_target_layer(score.astype(np.float32), gt, im_info, [16], anchors, A, rng=rng)
        pt = ora.proposal_target_layer(rois_all, sc_all, gt, c["classes"], rng=rng, batch_size=c["batch"], bg_lo=0.0)
        rois, roi_scores, labels, tg, iw, ow = pt
        print("proposals %d -> %d sampled (%d fg); rpn labels: %d fg %d bg  (%.1fs)" % (rois_all.shape[0], rois.shape[0], int((labels > 0).sum()),
              int((at[0] == 1).sum()), int((at[0] == 0).sum()), time.time() - t0), flush=True)
        feat_nhwc = feat.permute(0, 2, 3, 1).contiguous().numpy()
        pool5 = ora.crop_and_resize(feat_nhwc[0].astype(np.float32), rois.astype(np.float32), 16.0, 7)
        fc7 = ref.tail(pool5)
        cls_score, _, bbox_pred = ref.classify(fc7, test_mode=False)
        s_c = np.float32(2.0 / max(float(np.std(cls_score)), 1e-12))
        s_b = np.float32(1.0 / max(float(np.std(bbox_pred)), 1e-12))       # normalised targets are O(1) (stds 0.1 / 0.2)
        fx.update(cls_scale=s_c, bbox_scale=s_b)
        for name, sc_ in (("/cls_score/weights", s_c), ("/bbox_pred/weights", s_b)):
            v[scope + name] = (v[scope + name] * sc_).astype(np.float32)
    names = ref.bn_order
    fx["bn_names"] = np.array(names)
    fx["bn_mean"] = np.concatenate([v[s + "/BatchNorm/moving_mean"] for s in names]).astype(np.float32)
    fx["bn_var"] = np.concatenate([v[s + "/BatchNorm/moving_variance"] for s in names]).astype(np.float32)
    at_d = dict(rpn_labels=at[0], rpn_bbox_targets=at[1], rpn_bbox_inside_weights=at[2], rpn_bbox_outside_weights=at[3])
    pt_d = dict(labels=labels, bbox_targets=tg, bbox_inside_weights=iw, bbox_outside_weights=ow)
    fx.update(gt=gt, rois=rois.astype(np.float32), roi_scores=roi_scores.astype(np.float32).reshape(-1),
              **{"at_" + k: np.asarray(a, dtype=np.float32) for k, a in at_d.items()},
              **{"pt_" + k: np.asarray(a, dtype=np.float32) for k, a in pt_d.items()})
    del ref
    # ---- pass 2: float64 autograd of the reference graph on those constants; pass 3: the float32 control
    trainable = net.trainable_scope
    results = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t1 = time.time()
        tr = TrainRef(v, c["layers"], c["classes"], c["scales"], c["ratios"], trainable, dtype=dtype)
        losses = tr.losses(image, rois, at_d, pt_d)
        total = sum(losses.values())
        total.backward()
        results[tag] = (dict((k, float(x.item())) for k, x in losses.items()), collect(tr, scope, fs.TRAIN_GRAD_SCOPES))
        print("%s: %s  (%.1fs)" % (tag, "  ".join("%s %.6f" % kv for kv in results[tag][0].items()), time.time() - t1), flush=True)
        del tr, losses, total
    # ReLU gate margins of the RPN 3x3 layer.  Its upstream gradient is SPARSE: only the <= 256 sampled anchors' pixels carry any
    # (network.py:282-297), so one gate of relu(conv + b) that flips at an active pixel -- a pre-activation within rounding of zero
    # -- moves that output channel's bias / filter gradient by ~1/300 of its size.  Stored: per output channel, the smallest
    # |pre-activation| over the active pixels (float64), so the test can tell a flipped gate from an arithmetic error.
    with torch.no_grad():
        r64 = DenseRef(v, c["layers"], c["classes"], c["scales"], c["ratios"], dtype=torch.float64)
        feat = r64.head(image)
        pre = torch.nn.functional.conv2d(feat, r64.w(scope + "/rpn_conv/3x3/weights"), r64.w(scope + "/rpn_conv/3x3/biases"), padding=1)[0].numpy()
    lab = np.asarray(at[0]).reshape(A, pre.shape[1], pre.shape[2])
    active = (lab >= 0).any(axis=0)                                        # [H, W]
    fx["rpn_gate_margin"] = np.abs(pre[:, active]).min(axis=1).astype(np.float32)
    fx["rpn_pre_absmax"] = np.float64(np.abs(pre).max())
    print("rpn gates: %d active pixels, pre |max| %.3g, channels with a gate within 2e-4 of zero: %d" % (
        int(active.sum()), float(fx["rpn_pre_absmax"]), int((fx["rpn_gate_margin"] < 2e-4 * float(fx["rpn_pre_absmax"])).sum())), flush=True)
    l64, g64 = results["f64"]
    l32, g32 = results["f32"]
    for k in fs.LOSS_KEYS:
        fx["loss_" + k] = np.float64(l64[k])
        fx["ctrl_loss_" + k] = np.float64(l32[k])
    for key in fs.TRAIN_GRAD_SCOPES:
        fx["grad_" + key] = g64[key][0].astype(np.float32)
        fx["gabs_" + key] = np.float64(g64[key][1])
        fx["ctrl_grad_" + key] = g32[key][0].astype(np.float32)
        err = float(np.abs(g32[key][0].astype(np.float64) - g64[key][0]).max()) / max(g64[key][1], 1e-300)
        fx["ctrl_gerr_" + key] = np.float64(err)
        print("  grad %-28s |g|max %.3e  f32 control vs f64 %.2e  (%d samples)" % (key, g64[key][1], err, g64[key][0].size))
    os.makedirs(fs.GOLD, exist_ok=True)
    path = fs.train_fixture_path("c5")
    np.savez_compressed(path, **fx)
    print("wrote %s (%.2f MB) in %.1fs" % (path, os.path.getsize(path) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
