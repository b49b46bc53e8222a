#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/full_<config>_<weights>.npz: the float64 end-to-end reference of
BASELINE.json configs[1] (c2) / configs[2] (c3) on the seeded weights + image of oracle/fullsize.py.

    python oracle/gen_fullsize.py --config c2 --weights damped
    python oracle/gen_fullsize.py --config c2 --weights calibrated
    python oracle/gen_fullsize.py --config c3 --weights calibrated

Runs on host cores only (minutes); the GPU tests (tests/test_fullsize_gpu.py) rebuild the same weights and image from
the seeds, apply the fixture's RPN head scales / batch-norm statistics and compare.  Dense arithmetic: oracle/dense_ref.py
in float64 (PARITY UNPINNED for the TF/slim semantics, see that file); proposal / crop / per-class stages: the pinned
numpy/C oracle, fed with the float32 casts of the float64 tensors (what the reference's py_func seams would see).
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import frcnn_oracle as ora  # noqa: E402
import fullsize as fs  # noqa: E402


def calib_class(Base):
    """Base (a dense_ref restatement class) whose bn() can (re)define the frozen statistics from the tensor it normalises:
    moving_mean / moving_variance := per-channel mean / variance of the convolution output, rounded to float32 BEFORE use (the GPU
    loads the f32 values)."""
    class CalibRef(Base):
        calibrate = False

        def bn(self, x, scope, eps=1e-5):
            if self.calibrate:
                mean = x.mean(dim=(0, 2, 3)).numpy().astype(np.float32)
                var = x.var(dim=(0, 2, 3), unbiased=False).numpy().astype(np.float32)
                var = np.maximum(var, np.float32(1e-6))
                self.v[scope + "/BatchNorm/moving_mean"] = mean
                self.v[scope + "/BatchNorm/moving_variance"] = var
                self._cache.pop(scope + "/BatchNorm/moving_mean", None)
                self._cache.pop(scope + "/BatchNorm/moving_variance", None)
                self.bn_order.append(scope)
            return Base.bn(self, x, scope, eps)
    return CalibRef


def control_pass(c, v, image, rois, ref64=None):
    """The float32 control: the same restatement run by plain torch-CPU float32 on the same weights; the tail sees the f32 head
    (as on the device), cropped at the reference's rois.  Returns (tensors for full_*_ctrl.npz, scalar errors vs float64 or {})."""
    c32 = fs.make_ref(c, v, torch.float32)
    with torch.no_grad():
        feat32 = c32.head(image)
        s32, p32, b32 = c32.rpn(feat32)
        h32 = feat32.permute(0, 2, 3, 1).contiguous().numpy()
        fc32 = c32.tail(ora.crop_and_resize(h32[0], rois.astype(np.float32), 16.0, 7, max_pool=c32.max_pool_crop))
        cs32, cp32, bp32 = c32.classify(fc32)
    f = lambda a: np.asarray(a).astype(np.float32)
    tensors = dict(head_sub=f(h32[0, ::4, ::4, :]), rpn_cls_score=f(s32), rpn_cls_prob=f(p32), rpn_bbox_pred=f(b32),
                   fc7_sub=f(fc32.numpy()[::8]), cls_score=f(cs32), cls_prob=f(cp32), bbox_pred=f(bp32))
    errs = {}
    if ref64 is not None:
        errs = dict(ctrl_head=fs.rel_err(h32, ref64["head"]), ctrl_rpn_cls_score=fs.rel_err(s32, ref64["score"]),
                    ctrl_rpn_cls_prob=fs.rel_err(p32, ref64["prob"]), ctrl_rpn_bbox_pred=fs.rel_err(b32, ref64["bbox"]),
                    ctrl_cls_score=fs.rel_err(cs32, ref64["cls_score"]), ctrl_cls_prob_abs=float(np.abs(cp32 - ref64["cls_prob"]).max()),
                    ctrl_bbox_pred=fs.rel_err(bp32, ref64["bbox_pred"]))
    return tensors, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(fs.CONFIGS), default="c2")
    ap.add_argument("--weights", choices=["damped", "calibrated"], default="damped")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--ctrl-only", action="store_true", help="only (re)write full_<config>_<weights>_ctrl.npz: the float32 control's "
                    "tensors, from the committed fixture's weights and rois (seconds; no float64 pass)")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    c = fs.CONFIGS[args.config]
    t0 = time.time()
    if args.ctrl_only:
        fx = np.load(fs.fixture_path(args.config, args.weights))
        net, v = fs.base_variables(args.config, args.weights)
        fs.apply_fixture(v, net._scope, fx)
        tensors, _ = control_pass(c, v, fs.synth_image(c), fx["rois"])
        np.savez_compressed(fs.ctrl_path(args.config, args.weights), **tensors)
        print("wrote %s (%.2f MB) in %.1fs" % (fs.ctrl_path(args.config, args.weights), os.path.getsize(fs.ctrl_path(args.config, args.weights)) / 1e6,
                                               time.time() - t0))
        return
    net, v = fs.base_variables(args.config, args.weights)
    scope = net._scope
    image = fs.synth_image(c)
    im_info = np.array([c["H"], c["W"], c["scale"]], dtype=np.float32)
    ref = fs.make_ref(c, v, torch.float64, cls=calib_class(fs.ref_class(c)))
    ref.bn_order = []
    ref.calibrate = args.weights == "calibrated"
    A = ref.A
    with torch.no_grad():
        feat = ref.head(image)
        print("head %.1fs  |x|max %.3g  std %.3g" % (time.time() - t0, float(feat.abs().max()), float(feat.std())), flush=True)
        score, prob, bbox = ref.rpn(feat)
        # RPN head scales: logit spread ~1 (pair difference), deltas ~N(0, 0.2^2); f32 constants, then recompute
        s_cls = np.float32(1.0 / max(float(np.std(score)), 1e-12))
        s_box = np.float32(0.2 / max(float(np.std(bbox)), 1e-12))
        fx = dict(rpn_cls_scale=s_cls, rpn_box_scale=s_box)
        v[scope + "/rpn_cls_score/weights"] = (v[scope + "/rpn_cls_score/weights"] * s_cls).astype(np.float32)
        v[scope + "/rpn_bbox_pred/weights"] = (v[scope + "/rpn_bbox_pred/weights"] * s_box).astype(np.float32)
        ref._cache.pop(scope + "/rpn_cls_score/weights", None)
        ref._cache.pop(scope + "/rpn_bbox_pred/weights", None)
        score, prob, bbox = ref.rpn(feat)
        print("rpn %.1fs  score std %.3g  bbox std %.3g" % (time.time() - t0, float(np.std(score)), float(np.std(bbox))), flush=True)
        H, W = feat.shape[2], feat.shape[3]
        anchors, _ = ora.generate_anchors_pre(H, W, 16, c["scales"], c["ratios"])
        rois, roi_scores = ora.proposal_layer(prob.astype(np.float32), bbox.astype(np.float32), im_info, "TEST", [16], anchors, A,
                                              pre_nms_topN=c["pre"], post_nms_topN=c["post"], nms_thresh=0.7)
        print("proposals: %d" % rois.shape[0], flush=True)
        feat_nhwc = feat.permute(0, 2, 3, 1).contiguous().numpy()
        pool5 = ora.crop_and_resize(feat_nhwc[0].astype(np.float32), rois.astype(np.float32), 16.0, 7, max_pool=ref.max_pool_crop)
        fc7 = ref.tail(pool5)
        print("tail %.1fs  fc7 |x|max %.3g" % (time.time() - t0, float(fc7.abs().max())), flush=True)
        cls_score, cls_prob, bbox_pred = ref.classify(fc7)
        # class-head scales (SURVEY 8d "per-class stage": logits ~N(0, 2^2), de-normalised deltas ~N(0, 0.1^2)): random-init heads
        # give near-uniform probabilities, i.e. a per-class stage full of ties; f32 constants, then recompute
        s_c = np.float32(2.0 / max(float(np.std(cls_score)), 1e-12))
        s_b = np.float32(0.1 / max(float(np.std(bbox_pred)), 1e-12))
        fx.update(cls_scale=s_c, bbox_scale=s_b)
        for name, sc in (("/cls_score/weights", s_c), ("/bbox_pred/weights", s_b)):
            v[scope + name] = (v[scope + name] * sc).astype(np.float32)
            ref._cache.pop(scope + name, None)
        cls_score, cls_prob, bbox_pred = ref.classify(fc7)
    orig = (int(c["H"] / c["scale"]), int(c["W"] / c["scale"]), 3)
    sc, boxes = ora.im_detect_post(cls_prob.astype(np.float32), bbox_pred.astype(np.float32), rois.astype(np.float32), float(c["scale"]), orig)
    dets = ora.detections_to_records(ora.test_net_post(sc, boxes, c["classes"], max_per_image=c["max_per_image"]))
    print("detections: %d  cls_prob max %.4f" % (dets.shape[0], float(cls_prob[:, 1:].max())), flush=True)
    f = lambda a: np.asarray(a).astype(np.float32)      # float32 casts of the float64 results: what the reference's f32 graph hands on
    fx.update(rpn_cls_score=f(score), rpn_cls_prob=f(prob), rpn_bbox_pred=f(bbox), rois=rois.astype(np.float32),
              roi_scores=roi_scores.astype(np.float32), cls_score=f(cls_score), cls_prob=f(cls_prob), bbox_pred=f(bbox_pred),
              dets=dets.astype(np.float32), head_sub=feat_nhwc[0, ::4, ::4, :].astype(np.float32),
              head_absmax=np.float64(np.abs(feat_nhwc).max()), fc7_sub=fc7.numpy()[::8].astype(np.float32),
              fc7_absmax=np.float64(fc7.abs().max()))
    # control: what plain float32 arithmetic (torch-CPU, the same restatement) loses against float64 on THIS network --
    # the yardstick for "as exact as f32 gets" when the 1e-4 budget is smaller than f32's own noise on a graph
    ref.calibrate = False
    tensors, errs = control_pass(c, v, image, rois, dict(head=feat_nhwc, score=score, prob=prob, bbox=bbox, cls_score=cls_score, cls_prob=cls_prob,
                                                           bbox_pred=bbox_pred))
    fx.update(errs)
    np.savez_compressed(fs.ctrl_path(args.config, args.weights), **tensors)
    print("f32 control vs f64: " + "  ".join("%s %.2e" % (k[5:], float(fx[k])) for k in sorted(fx) if k.startswith("ctrl_")), flush=True)
    if args.weights == "calibrated":
        names = ref.bn_order
        fx["bn_names"] = np.array(names)
        fx["bn_mean"] = np.concatenate([v[s + "/BatchNorm/moving_mean"] for s in names]).astype(np.float32)
        fx["bn_var"] = np.concatenate([v[s + "/BatchNorm/moving_variance"] for s in names]).astype(np.float32)
    os.makedirs(fs.GOLD, exist_ok=True)
    path = fs.fixture_path(args.config, args.weights)
    np.savez_compressed(path, **fx)
    print("wrote %s (%.2f MB) in %.1fs" % (path, os.path.getsize(path) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
