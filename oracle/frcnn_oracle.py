"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy + oracle/oracle_c.c) of the reference's
numpy/Cython detection path.  Never imported by the product path (tf-faster-rcnn_amd/); only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

PINNED: every function here is checked against the reference's own code, imported from
/root/reference through oracle/ref_shim.py, by oracle/gen_golden.py (which also writes the golden
fixtures under tests/golden/) and by tests/test_oracle_golden.py.

All file:line citations are relative to /root/reference/lib.  Tie policy of this oracle:
sorting is (score descending, index ascending) -- the reference's `argsort()[::-1]` is an unstable
sort whose tie order is CPU-ISA dependent, so bit-exactness is only claimed
on tie-free scores.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
f32 = np.float32


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_cpu_nms.restype = ctypes.c_int
        _LIB.oracle_bbox_overlaps.restype = None
        _LIB.oracle_crop_and_resize.restype = None
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------- anchors
def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """layer_utils/generate_anchors.py:41-105.  float64; np.round = half-to-even (lines 90-91).
    Output order: ratio-major, scale-minor (vstack over ratio anchors, :50-51)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    w = h = float(base_size)                      # base window (0,0,15,15): w = 15-0+1
    cx = cy = 0.5 * (base_size - 1)               # x_ctr = x1 + 0.5*(w-1)
    ws = np.round(np.sqrt(w * h / ratios))        # _ratio_enum :87-92
    hs = np.round(ws * ratios)
    out = []
    for rw, rh in zip(ws, hs):                    # _scale_enum :96-105 on each ratio anchor
        # the ratio anchor is (cx-0.5(rw-1), ..); its _whctrs gives back rw, rh, cx, cy exactly
        x1, y1 = cx - 0.5 * (rw - 1), cy - 0.5 * (rh - 1)
        x2, y2 = cx + 0.5 * (rw - 1), cy + 0.5 * (rh - 1)
        aw, ah = x2 - x1 + 1, y2 - y1 + 1
        acx, acy = x1 + 0.5 * (aw - 1), y1 + 0.5 * (ah - 1)
        for s in scales:
            sw, sh = aw * s, ah * s
            out.append([acx - 0.5 * (sw - 1), acy - 0.5 * (sh - 1),
                        acx + 0.5 * (sw - 1), acy + 0.5 * (sh - 1)])
    return np.array(out, dtype=np.float64)


def generate_anchors_pre(height, width, feat_stride, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
    """layer_utils/snippets.py:14-30.  anchors f32 [H*W*A, 4], index (y*W + x)*A + a."""
    base = generate_anchors(ratios=anchor_ratios, scales=anchor_scales)
    A = base.shape[0]
    sx = np.arange(width, dtype=np.int64) * int(feat_stride)
    sy = np.arange(height, dtype=np.int64) * int(feat_stride)
    shifts = np.stack([np.tile(sx, height), np.repeat(sy, width),
                       np.tile(sx, height), np.repeat(sy, width)], axis=1)       # [K,4], x fastest
    anchors = (shifts[:, None, :] + base[None, :, :]).reshape(-1, 4).astype(f32)
    return anchors, np.int32(anchors.shape[0])


# ----------------------------------------------------------------------------- box codec
def bbox_transform_inv(boxes, deltas):
    """model/bbox_transform.py:35-65.  f32 throughout; deltas [N, 4*k] strided 0::4."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    b = boxes.astype(deltas.dtype, copy=False)
    one, half = deltas.dtype.type(1.0), deltas.dtype.type(0.5)
    w = (b[:, 2] - b[:, 0]) + one
    h = (b[:, 3] - b[:, 1]) + one
    cx = b[:, 0] + half * w
    cy = b[:, 1] + half * h
    d = deltas.reshape(deltas.shape[0], -1, 4)
    pcx = d[:, :, 0] * w[:, None] + cx[:, None]
    pcy = d[:, :, 1] * h[:, None] + cy[:, None]
    pw = np.exp(d[:, :, 2]) * w[:, None]
    ph = np.exp(d[:, :, 3]) * h[:, None]
    out = np.empty_like(d)
    out[:, :, 0] = pcx - half * pw
    out[:, :, 1] = pcy - half * ph
    out[:, :, 2] = pcx + half * pw
    out[:, :, 3] = pcy + half * ph
    return out.reshape(deltas.shape)


def bbox_transform(ex_rois, gt_rois):
    """model/bbox_transform.py:14-32 (encode)."""
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    ecx = ex_rois[:, 0] + 0.5 * ew
    ecy = ex_rois[:, 1] + 0.5 * eh
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    gcx = gt_rois[:, 0] + 0.5 * gw
    gcy = gt_rois[:, 1] + 0.5 * gh
    return np.stack([(gcx - ecx) / ew, (gcy - ecy) / eh, np.log(gw / ew), np.log(gh / eh)], axis=1)


def clip_boxes(boxes, im_shape):
    """model/bbox_transform.py:68-81.  All four coords clamped to [0, dim-1]; in place."""
    hi_x = boxes.dtype.type(im_shape[1]) - boxes.dtype.type(1)
    hi_y = boxes.dtype.type(im_shape[0]) - boxes.dtype.type(1)
    v = boxes.reshape(boxes.shape[0], -1, 4)
    for c, hi in ((0, hi_x), (1, hi_y), (2, hi_x), (3, hi_y)):
        v[:, :, c] = np.maximum(np.minimum(v[:, :, c], hi), 0)
    return boxes


def clip_boxes_final(boxes, im_shape):
    """model/test.py:67-77 (_clip_boxes): x1,y1 >= 0 ; x2 <= W-1 ; y2 <= H-1 (ORIGINAL image)."""
    v = boxes.reshape(boxes.shape[0], -1, 4)
    v[:, :, 0] = np.maximum(v[:, :, 0], 0)
    v[:, :, 1] = np.maximum(v[:, :, 1], 0)
    v[:, :, 2] = np.minimum(v[:, :, 2], im_shape[1] - 1)
    v[:, :, 3] = np.minimum(v[:, :, 3], im_shape[0] - 1)
    return boxes


# ----------------------------------------------------------------------------- NMS / IoU (C)
def order_desc(scores):
    """(score desc, index asc): the deterministic stand-in for `scores.argsort()[::-1]`."""
    return np.argsort(-scores.astype(np.float64), kind="stable")


def cpu_nms(dets, thresh):
    """nms/cpu_nms.pyx:17-68 restated in C (oracle_c.c:oracle_cpu_nms).  dets f32 [K,5]; returns
    kept ORIGINAL indices in score order; suppress iff (double)ovr >= thresh (cpu_nms.c:2239-2241)."""
    dets = np.ascontiguousarray(dets, dtype=f32)
    k = dets.shape[0]
    if k == 0:
        return []
    order = order_desc(dets[:, 4]).astype(np.int64)
    keep = np.empty(k, dtype=np.int64)
    n = _lib().oracle_cpu_nms(_p(dets), ctypes.c_int(k), _p(order), ctypes.c_double(float(thresh)), _p(keep))
    return keep[:n].tolist()


def gpu_nms(dets, thresh):
    """The reference's OTHER suppression rule: nms/nms_kernel.cu:24-33,71 (`devIoU(...) > nms_overlap_thresh`, all float32,
    threshold a C float) == nms/py_cpu_nms.py:10-38 (`ovr <= thresh` kept; float32 arrays against a weak Python scalar).
    Same +1 areas and operation order as cpu_nms; only the comparison differs.  PINNED against the reference's own
    py_cpu_nms (oracle/gen_golden.py); the CUDA kernel itself cannot run here (nvcc may also contract `Sa + Sb` into an FMA,
    which no CPU statement of the reference does)."""
    d = np.ascontiguousarray(dets, dtype=f32)
    k = d.shape[0]
    if k == 0:
        return []
    x1, y1, x2, y2 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    areas = ((x2 - x1) + f32(1)) * ((y2 - y1) + f32(1))
    order = order_desc(d[:, 4])
    thr = f32(thresh)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(f32(0), (np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest])) + f32(1))
        h = np.maximum(f32(0), (np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest])) + f32(1))
        inter = w * h
        ovr = inter / ((areas[i] + areas[rest]) - inter)
        order = rest[ovr <= thr]
    return keep


def nms(dets, thresh, force_cpu=False):
    """model/nms_wrapper.py:15-23 with cfg.USE_GPU_NMS = False."""
    if dets.shape[0] == 0:
        return []
    return cpu_nms(dets, thresh)


def bbox_overlaps(boxes, query_boxes):
    """utils/bbox.pyx:15-55: IoU with the +1 convention, float64 [N,K]."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    q = np.ascontiguousarray(query_boxes, dtype=np.float64)
    out = np.zeros((boxes.shape[0], q.shape[0]), dtype=np.float64)
    if out.size:
        _lib().oracle_bbox_overlaps(_p(boxes), ctypes.c_int(boxes.shape[0]), _p(q),
                                    ctypes.c_int(q.shape[0]), ctypes.c_int(q.shape[1]), _p(out))
    return out


# ----------------------------------------------------------------------------- proposal layers
def proposal_layer(rpn_cls_prob, rpn_bbox_pred, im_info, cfg_key, _feat_stride, anchors, num_anchors,
                   pre_nms_topN=None, post_nms_topN=None, nms_thresh=None):
    """layer_utils/proposal_layer.py:16-53.  cfg values default to model/config.py:142-148,192-198."""
    if isinstance(cfg_key, bytes):
        cfg_key = cfg_key.decode("utf-8")
    dflt = {"TEST": (6000, 300, 0.7), "TRAIN": (12000, 2000, 0.7)}[cfg_key]
    pre_n = dflt[0] if pre_nms_topN is None else pre_nms_topN
    post_n = dflt[1] if post_nms_topN is None else post_nms_topN
    thr = dflt[2] if nms_thresh is None else nms_thresh
    scores = rpn_cls_prob[:, :, :, num_anchors:].reshape(-1)           # fg = channels [A:2A]  (:27)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    props = clip_boxes(bbox_transform_inv(anchors, deltas), im_info[:2])  # (:30-31)
    order = order_desc(scores)                                           # (:34)
    if pre_n > 0:
        order = order[:pre_n]
    props, scores = props[order], scores[order].reshape(-1, 1)
    keep = nms(np.hstack((props, scores)), thr)                          # (:41)
    if post_n > 0:
        keep = keep[:post_n]
    props, scores = props[keep], scores[keep]
    blob = np.hstack((np.zeros((props.shape[0], 1), dtype=f32), props.astype(f32, copy=False)))
    return blob, scores


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, im_info, _feat_stride, anchors, num_anchors, rpn_top_n=5000):
    """layer_utils/proposal_top_layer.py:17-55 for the (deterministic) case length >= rpn_top_n."""
    scores = rpn_cls_prob[:, :, :, num_anchors:].reshape(-1)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    assert scores.shape[0] >= rpn_top_n, "random-fill branch (:30-33) is RNG dependent; not restated"
    top = order_desc(scores)[:rpn_top_n]
    props = clip_boxes(bbox_transform_inv(anchors[top], deltas[top]), im_info[:2])
    blob = np.hstack((np.zeros((props.shape[0], 1), dtype=f32), props.astype(f32, copy=False)))
    return blob, scores[top].reshape(-1, 1)


# ----------------------------------------------------------------------------- USE_E2E_TF graph (config.py:275)
# The reference's default graph replaces the numpy layers by TensorFlow ops.  The graph wiring below is PINNED by running
# the reference's own *_tf function bodies on oracle/tf_numpy_shim.py (gen_golden.py).  The two TensorFlow kernels are
# third-party code absent from /root/reference -- restated from TensorFlow r1.2 (the version README.md:56 names),
# tensorflow/core/kernels/non_max_suppression_op.cc and topk_op.cc: PARITY UNPINNED for these two functions.
def tf_non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """non_max_suppression_op.cc (r1.2): candidates in decreasing score order (std::sort on score; ties are
    implementation-defined there, (score desc, index asc) here); a candidate is selected unless its IoU with an already
    selected box is > iou_threshold; stops at max_output_size.  ComputeIOU: corners normalised with min/max, area WITHOUT
    +1, 0 when either area <= 0, all in float32.  Returns int32 indices."""
    b = np.ascontiguousarray(boxes, dtype=f32)
    sc = np.asarray(scores, dtype=f32).reshape(-1)
    thr = f32(iou_threshold)
    y0, y1 = np.minimum(b[:, 0], b[:, 2]), np.maximum(b[:, 0], b[:, 2])
    x0, x1 = np.minimum(b[:, 1], b[:, 3]), np.maximum(b[:, 1], b[:, 3])
    area = (y1 - y0) * (x1 - x0)
    order = order_desc(sc)
    sel = np.empty((max(int(max_output_size), 0),), dtype=np.int64)
    n = 0
    for i in order:
        if n >= max_output_size:
            break
        if n and area[i] > 0:
            s_ = sel[:n]
            ih = np.maximum(np.minimum(y1[i], y1[s_]) - np.maximum(y0[i], y0[s_]), f32(0))
            iw = np.maximum(np.minimum(x1[i], x1[s_]) - np.maximum(x0[i], x0[s_]), f32(0))
            inter = ih * iw
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = inter / ((area[i] + area[s_]) - inter)
            if np.any((iou > thr) & (area[s_] > 0)):
                continue
        sel[n] = i
        n += 1
    return sel[:n].astype(np.int32)


def tf_top_k(values, k):
    """topk_op.cc: the k largest, descending; equal values -> lower index first.  Returns (values, int32 indices)."""
    v = np.asarray(values)
    idx = order_desc(v.reshape(-1))[:k]
    return v.reshape(-1)[idx], idx.astype(np.int32)


def generate_anchors_pre_tf(height, width, feat_stride=16, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
    """snippets.py:32-49: base anchors TRUNCATED to int32 (tf.constant(..., dtype=tf.int32)), integer shifts, then f32."""
    base = generate_anchors(ratios=np.array(anchor_ratios), scales=np.array(anchor_scales)).astype(np.int32)
    A = base.shape[0]
    xs = np.arange(int(width), dtype=np.int32) * np.int32(feat_stride)
    ys = np.arange(int(height), dtype=np.int32) * np.int32(feat_stride)
    out = np.empty((int(height), int(width), A, 4), dtype=np.int32)
    out[..., 0] = base[None, None, :, 0] + xs[None, :, None]
    out[..., 1] = base[None, None, :, 1] + ys[:, None, None]
    out[..., 2] = base[None, None, :, 2] + xs[None, :, None]
    out[..., 3] = base[None, None, :, 3] + ys[:, None, None]
    n = int(height) * int(width) * A
    return out.reshape(n, 4).astype(f32), n


def proposal_layer_tf(rpn_cls_prob, rpn_bbox_pred, im_info, anchors, num_anchors, post_nms_topN=300, nms_thresh=0.7):
    """proposal_layer.py:56-84: decode + clip ALL anchors, TF NMS over all of them, at most post_nms_topN rows."""
    scores = np.ascontiguousarray(rpn_cls_prob[:, :, :, num_anchors:]).reshape(-1)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    props = clip_boxes(bbox_transform_inv(anchors.astype(f32), deltas), im_info[:2])
    keep = tf_non_max_suppression(props, scores, post_nms_topN, nms_thresh)
    blob = np.concatenate([np.zeros((keep.shape[0], 1), dtype=f32), props[keep].astype(f32)], axis=1)
    return blob, scores[keep].reshape(-1, 1)


def proposal_top_layer_tf(rpn_cls_prob, rpn_bbox_pred, im_info, anchors, num_anchors, rpn_top_n=5000):
    """proposal_top_layer.py:58-85: tf.nn.top_k, decode + clip only those, no NMS."""
    scores = np.ascontiguousarray(rpn_cls_prob[:, :, :, num_anchors:]).reshape(-1)
    top_scores, top = tf_top_k(scores, rpn_top_n)
    props = clip_boxes(bbox_transform_inv(anchors[top].astype(f32), rpn_bbox_pred.reshape(-1, 4)[top]), im_info[:2])
    blob = np.concatenate([np.zeros((rpn_top_n, 1), dtype=f32), props.astype(f32)], axis=1)
    return blob, top_scores.reshape(-1, 1)


# ----------------------------------------------------------------------------- image preprocessing (model/test.py:26-58)
# cv2.resize is third-party (OpenCV, not vendored, cv2 not installable offline): restated from OpenCV 3.x
# modules/imgproc/src/resize.cpp (32f INTER_LINEAR path).  PARITY UNPINNED for cv2_resize_linear; the scale rule and
# the mean subtraction around it are the reference's own lines and are restated literally.
def cv2_resize_linear(im, fx, fy):
    """cv2.resize(im, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for a float32 [h,w,c] image."""
    im = np.ascontiguousarray(im, dtype=f32)
    h, w = im.shape[:2]
    OW, OH = int(np.round(w * fx)), int(np.round(h * fy))           # saturate_cast<int> == round half to even
    sx_inv, sy_inv = 1.0 / fx, 1.0 / fy

    def taps(n_out, inv, n_in, clamp_weight):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * inv - 0.5).astype(f32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(f32)).astype(f32)
        if clamp_weight:                                              # columns: weight zeroed at the borders
            lo = s < 0
            f[lo], s[lo] = 0, 0
            last = s + 1 >= n_in
            hi = s >= n_in - 1
            f[hi], s[hi] = 0, n_in - 1
            return s, np.where(last, s, s + 1), f, last
        return np.clip(s, 0, n_in - 1), np.clip(s + 1, 0, n_in - 1), f, None     # rows: indices clamped, weight kept

    x0, x1, ax, last = taps(OW, sx_inv, w, True)
    y0, y1, ay, _ = taps(OH, sy_inv, h, False)
    a0, a1 = (f32(1) - ax)[None, :, None], ax[None, :, None]
    rows0, rows1 = im[y0], im[y1]
    one = last[None, :, None]
    r0 = np.where(one, rows0[:, x0] * f32(1), rows0[:, x0] * a0 + rows0[:, x1] * a1).astype(f32)
    r1 = np.where(one, rows1[:, x0] * f32(1), rows1[:, x0] * a0 + rows1[:, x1] * a1).astype(f32)
    b0, b1 = (f32(1) - ay)[:, None, None], ay[:, None, None]
    return (r0 * b0 + r1 * b1).astype(f32)


def get_image_blob(im, pixel_means, target_size=600, max_size=1000):
    """model/test.py:26-58 for one scale: (blob [1,H,W,3] f32, im_scale)."""
    im_orig = im.astype(f32, copy=True)
    im_orig -= np.asarray(pixel_means)                                # float64 means: f64 subtraction, stored as f32 (:35-36)
    smin, smax = np.min(im_orig.shape[0:2]), np.max(im_orig.shape[0:2])
    im_scale = float(target_size) / float(smin)
    if np.round(im_scale * smax) > max_size:
        im_scale = float(max_size) / float(smax)
    return cv2_resize_linear(im_orig, im_scale, im_scale)[None], im_scale


# ----------------------------------------------------------------------------- test-time post-processing
def im_detect_post(scores, bbox_pred, rois, im_scale, im_shape):
    """model/test.py:95-102: rois/scale, per-class decode, final clip.  -> scores [R,C], boxes [R,4C]."""
    # rois f32 / im_scales[0] (an np.float64 scalar, test.py:58,95): under NEP-50 (numpy >= 2, the
    # numpy this reference runs on here) the division is carried out in float64 and
    # bbox_transform_inv casts back to f32 (bbox_transform.py:39).
    boxes = (rois[:, 1:5] / np.float64(im_scale)).astype(f32)
    pred = bbox_transform_inv(boxes, bbox_pred.reshape(bbox_pred.shape[0], -1).astype(f32))
    return scores.reshape(scores.shape[0], -1), clip_boxes_final(pred, im_shape)


def test_net_post(scores, boxes, num_classes, nms_thresh=0.3, max_per_image=100, thresh=0.0):
    """model/test.py:162-180: per-class `score > thresh`, NMS(TEST.NMS), then the global top-
    max_per_image score cut (`>= np.sort(all)[-max]`).  Returns list over classes of [n,5] f32."""
    out = [np.zeros((0, 5), dtype=f32)]
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > thresh)[0]
        dets = np.hstack((boxes[inds, j * 4:(j + 1) * 4], scores[inds, j][:, None])).astype(f32, copy=False)
        out.append(dets[nms(dets, nms_thresh), :])
    if max_per_image > 0:
        allsc = np.hstack([out[j][:, -1] for j in range(1, num_classes)])
        if len(allsc) > max_per_image:
            cut = np.sort(allsc)[-max_per_image]
            for j in range(1, num_classes):
                out[j] = out[j][out[j][:, -1] >= cut, :]
    return out


def detections_to_records(per_class):
    """Flatten test_net_post output to the device record layout [n,6] = x1,y1,x2,y2,score,cls
    (class-major, score-descending inside a class == the reference's all_boxes[j][i] order)."""
    rows = [np.hstack((d, np.full((d.shape[0], 1), j, dtype=f32))) for j, d in enumerate(per_class) if d.shape[0]]
    return np.vstack(rows).astype(f32) if rows else np.zeros((0, 6), dtype=f32)


# ----------------------------------------------------------------------------- RoI pooling
def crop_and_resize(feat, rois, feat_stride, pool, max_pool=False):
    """nets/resnet_v1.py:55-76 / nets/network.py:141-157 around tf.image.crop_and_resize (third
    party; TF kernel semantics restated in oracle_c.c -- PARITY UNPINNED).
    feat f32 [H,W,C] (NHWC, batch 1), rois f32 [R,5] in image coords -> [R,pool,pool,C].
    max_pool=True: crop at 2*pool then 2x2/2 max (network.py:152-157)."""
    feat = np.ascontiguousarray(feat, dtype=f32)
    rois = np.ascontiguousarray(rois, dtype=f32)
    H, W, C = feat.shape
    P = pool * 2 if max_pool else pool
    out = np.empty((rois.shape[0], P, P, C), dtype=f32)
    if out.size:
        _lib().oracle_crop_and_resize(_p(feat), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(C), _p(rois),
                                      ctypes.c_int(rois.shape[0]), ctypes.c_float(feat_stride),
                                      ctypes.c_int(P), _p(out))
    if max_pool:
        out = out.reshape(-1, pool, 2, pool, 2, C).max(axis=(2, 4))
    return out


# ----------------------------------------------------------------------------- training targets
def anchor_target_layer(rpn_cls_score, gt_boxes, im_info, _feat_stride, all_anchors, num_anchors,
                        rng=np.random, batchsize=256, fg_fraction=0.5, pos_ov=0.7, neg_ov=0.3, clobber_positives=False,
                        positive_weight=-1.0, inside_weights=(1.0, 1.0, 1.0, 1.0)):
    """layer_utils/anchor_target_layer.py:18-138; clobber_positives / positive_weight / inside_weights = TRAIN.RPN_CLOBBER_POSITIVES
    (:57-70), RPN_POSITIVE_WEIGHT (:96-109), RPN_BBOX_INSIDE_WEIGHTS (:91-93).  `rng` must expose numpy's legacy `choice` (the
    reference uses the global numpy.random state)."""
    A = num_anchors
    total = all_anchors.shape[0]
    height, width = rpn_cls_score.shape[1:3]
    inds_inside = np.where((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) &
                           (all_anchors[:, 2] < im_info[1]) & (all_anchors[:, 3] < im_info[0]))[0]     # :31-36
    anchors = all_anchors[inds_inside, :]
    labels = np.full((len(inds_inside),), -1, dtype=f32)
    ov = bbox_overlaps(anchors, gt_boxes[:, :4])                                                         # :47-49
    argmax = ov.argmax(axis=1)
    maxov = ov[np.arange(len(inds_inside)), argmax]
    gt_max = ov[ov.argmax(axis=0), np.arange(ov.shape[1])]
    gt_argmax = np.where(ov == gt_max)[0]                                                                # :52-55
    if not clobber_positives:
        labels[maxov < neg_ov] = 0
    labels[gt_argmax] = 1
    labels[maxov >= pos_ov] = 1
    if clobber_positives:
        labels[maxov < neg_ov] = 0
    num_fg = int(fg_fraction * batchsize)
    fg = np.where(labels == 1)[0]
    if len(fg) > num_fg:
        labels[rng.choice(fg, size=(len(fg) - num_fg), replace=False)] = -1                              # :73-78
    num_bg = batchsize - np.sum(labels == 1)
    bg = np.where(labels == 0)[0]
    if len(bg) > num_bg:
        labels[rng.choice(bg, size=(len(bg) - num_bg), replace=False)] = -1                              # :81-86
    targets = bbox_transform(anchors, gt_boxes[argmax, :][:, :4]).astype(f32, copy=False)               # :88-89,155-162
    inside_w = np.zeros((len(inds_inside), 4), dtype=f32)
    inside_w[labels == 1, :] = np.array(inside_weights)
    outside_w = np.zeros((len(inds_inside), 4), dtype=f32)
    if positive_weight < 0:
        num_examples = np.sum(labels >= 0)
        pos_w = neg_w = np.ones((1, 4)) * 1.0 / num_examples
    else:
        pos_w = positive_weight / np.sum(labels == 1)
        neg_w = (1.0 - positive_weight) / np.sum(labels == 0)
    outside_w[labels == 1, :] = pos_w
    outside_w[labels == 0, :] = neg_w

    def unmap(data, fill):
        if data.ndim == 1:
            ret = np.full((total,), fill, dtype=f32)
            ret[inds_inside] = data
        else:
            ret = np.full((total,) + data.shape[1:], fill, dtype=f32)
            ret[inds_inside, :] = data
        return ret
    labels = unmap(labels, -1).reshape((1, height, width, A)).transpose(0, 3, 1, 2).reshape((1, 1, A * height, width))
    targets = unmap(targets, 0).reshape((1, height, width, A * 4))
    inside_w = unmap(inside_w, 0).reshape((1, height, width, A * 4))
    outside_w = unmap(outside_w, 0).reshape((1, height, width, A * 4))
    return labels, targets, inside_w, outside_w


def proposal_target_layer(rpn_rois, rpn_scores, gt_boxes, num_classes, rng=np.random, batch_size=256,
                          fg_fraction=0.25, fg_thresh=0.5, bg_hi=0.5, bg_lo=0.0,
                          means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2), use_gt=False, inside_weights=(1.0, 1.0, 1.0, 1.0)):
    """layer_utils/proposal_target_layer.py:18-152 (IMS_PER_BATCH 1); use_gt = TRAIN.USE_GT (:30-36), inside_weights =
    TRAIN.BBOX_INSIDE_WEIGHTS (:78)."""
    if use_gt:
        zeros = np.zeros((gt_boxes.shape[0], 1), dtype=gt_boxes.dtype)
        rpn_rois = np.vstack((rpn_rois, np.hstack((zeros, gt_boxes[:, :-1]))))
        rpn_scores = np.vstack((rpn_scores, zeros))
    rois_per_image = batch_size
    fg_per_image = int(np.round(fg_fraction * rois_per_image))
    ov = bbox_overlaps(rpn_rois[:, 1:5], gt_boxes[:, :4])
    assign = ov.argmax(axis=1)
    maxov = ov.max(axis=1)
    labels = gt_boxes[assign, 4]
    fg = np.where(maxov >= fg_thresh)[0]
    bg = np.where((maxov < bg_hi) & (maxov >= bg_lo))[0]
    if fg.size > 0 and bg.size > 0:                                                                     # :119-135
        fg_per_image = min(fg_per_image, fg.size)
        fg = rng.choice(fg, size=int(fg_per_image), replace=False)
        bg_n = rois_per_image - fg_per_image
        bg = rng.choice(bg, size=int(bg_n), replace=bg.size < bg_n)
    elif fg.size > 0:
        fg = rng.choice(fg, size=int(rois_per_image), replace=fg.size < rois_per_image)
        fg_per_image = rois_per_image
    elif bg.size > 0:
        bg = rng.choice(bg, size=int(rois_per_image), replace=bg.size < rois_per_image)
        fg_per_image = 0
    else:
        raise RuntimeError("no fg and no bg rois (reference drops into pdb here, :133-135)")
    keep = np.append(fg, bg)
    labels = labels[keep]
    labels[int(fg_per_image):] = 0
    rois = rpn_rois[keep]
    roi_scores = rpn_scores[keep]
    t = bbox_transform(rois[:, 1:5], gt_boxes[assign[keep], :4])
    t = (t - np.array(means)) / np.array(stds)                                                          # :83-96
    data = np.hstack((labels[:, None], t)).astype(f32, copy=False)
    bbox_targets = np.zeros((labels.size, 4 * num_classes), dtype=f32)
    inside = np.zeros_like(bbox_targets)
    for i in np.where(labels > 0)[0]:                                                                    # :58-80
        c = int(4 * labels[i])
        bbox_targets[i, c:c + 4] = data[i, 1:]
        inside[i, c:c + 4] = inside_weights
    rois = rois.reshape(-1, 5)
    roi_scores = roi_scores.reshape(-1)
    labels = labels.reshape(-1, 1)
    outside = np.array(inside > 0).astype(f32)
    return rois, roi_scores, labels, bbox_targets, inside, outside
