"""Tensor-level wrappers over the C ABI.  torch is plumbing here: device buffers, streams.

Every function takes/returns torch CUDA tensors (float32, contiguous, NHWC) and launches on the
CURRENT torch stream (so torch.cuda.Event / torch.cuda.synchronize see the work).  Outputs may be
passed in (`out=`) so a captured hipGraph can reuse static buffers.
"""
import ctypes

import numpy as np
import torch

import frcnn_hip as _binding
from . import ACT_NONE, NMS_RULE_CPU, call, lib

_ws_cache = {}
_ws_retired = []
ws_scope = "default"     # set by callers that run several independent chains concurrently (one scope per stream)

# The per-SHAPE buffer set of the build that is running (Session.shape_scope; None outside one).  Everything whose size follows the image --
# named activation buffers, operand planes, arena results, scratch -- is registered there while it is set, so that evicting a shape's
# captured graph / recorded step frees (returns to torch's size-class pools) exactly what that shape needed.  Weight-shaped buffers stay
# session-wide: their producers run under `unscoped()`.
scope_store = None


class unscoped(object):
    """`with unscoped():` -- buffers allocated inside belong to the session, not to the image shape being built (filter images, solver state)."""

    def __enter__(self):
        global scope_store
        self.prev, scope_store = scope_store, None
        return self

    def __exit__(self, *exc):
        global scope_store
        scope_store = self.prev
        return False


stream_pin = None        # a ctypes stream handle: every launch of this module goes there instead of torch's current stream (see pinned_stream)


def _stream():
    return stream_pin if stream_pin is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class pinned_stream(object):
    """`with pinned_stream(s):` -- the launches of this module go to torch stream `s` without the per-call current_stream() lookup
    and without making `s` torch's current stream (a `with torch.cuda.stream(...)` costs ~10 us of host time per entry; the reverse
    sweep switches streams ~200 times a step).  Only for regions whose torch-level ops, if any, belong on torch's current stream."""

    def __init__(self, stream):
        self.handle = ctypes.c_void_p(stream.cuda_stream)

    def __enter__(self):
        global stream_pin
        self.prev, stream_pin = stream_pin, self.handle
        return self

    def __exit__(self, *exc):
        global stream_pin
        stream_pin = self.prev
        return False


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# ---- a training step as a replayable launch list (frcnn_hip/replay.py) ------------------------------------------------------------------
# `arena`: while set, the result tensors these wrappers allocate themselves are static session buffers (same address every step).
# `_binding.recorder`: while set, every call() and every stream-level operation below is appended to the recording of the step.
arena = None


def _empty(shape, dtype=torch.float32, device=None):
    if arena is not None:
        return arena.take(shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,), dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


def _zeros(shape, dtype=torch.float32, device=None):
    if arena is not None:
        return t_zero(arena.take(shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,), dtype, device))
    return torch.zeros(shape, dtype=dtype, device=device)


def t_zero(t):
    """t.zero_() on torch's current stream (recorded as an operation of the step)"""
    t.zero_()
    R = _binding.recorder
    if R is not None:
        R.add_op(lambda rec, t=t: t.zero_())
    return t


def t_copy(dst, src):
    """dst.copy_(src) on torch's current stream"""
    dst.copy_(src)
    R = _binding.recorder
    if R is not None:
        R.add_op(lambda rec, dst=dst, src=src: dst.copy_(src))
    return dst


def ev_record(ev, stream):
    ev.record(stream)
    R = _binding.recorder
    if R is not None:
        s = R.slot(stream)
        R.add_op(lambda rec, ev=ev, s=s: ev.record(rec.bound[s]))


def st_wait_event(stream, ev):
    stream.wait_event(ev)
    R = _binding.recorder
    if R is not None:
        s = R.slot(stream)
        R.add_op(lambda rec, ev=ev, s=s: rec.bound[s].wait_event(ev))


def st_wait_stream(stream, other):
    """stream.wait_stream(other)"""
    R = _binding.recorder
    if R is None:
        stream.wait_stream(other)
        return
    ev = torch.cuda.Event()                     # (torch's wait_stream makes a fresh event per call; a recording owns one per site)
    R.keep.append(ev)
    ev_record(ev, other)
    st_wait_event(stream, ev)


def host_op(fn):
    """fn(): host code that enqueues work through torch (a collective, a tensor expression): run now and, while a step is being
    recorded, again at every replay.  fn may take one argument -- None now, the Recording at replay (rec.bound[slot] is the stream
    bound to a slot that was looked up with recorder.slot(stream) at record time)."""
    import inspect
    takes = len(inspect.signature(fn).parameters) > 0
    out = fn(None) if takes else fn()
    R = _binding.recorder
    if R is not None:
        R.add_op(fn if takes else (lambda rec, fn=fn: fn()))
    return out


def _chk(t, dtype=torch.float32):
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise ValueError("expected a contiguous CUDA %s tensor, got %s %s contiguous=%s" %
                         (dtype, t.device, t.dtype, t.is_contiguous()))
    return t


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag); never reallocated inside a captured region
    as long as the first (warm-up) call already saw the largest request."""
    # inside a shape scope the scratch belongs to the shape (freed with it); a buffer a captured graph may still address is retired INTO the
    # same store, i.e. lives exactly as long as that graph
    store = _ws_cache if scope_store is None else scope_store
    key = ("ws", str(device), tag, ws_scope)
    buf = store.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            (_ws_retired if scope_store is None else store.setdefault(("ws_retired",), [])).append(buf)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        store[key] = buf
    return buf


# ------------------------------------------------------------------------------------------ anchors
def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """HOST float64 [A,4] base anchors (lib/layer_utils/generate_anchors.py:41-52)."""
    r = np.ascontiguousarray(ratios, dtype=np.float64)
    s = np.ascontiguousarray(scales, dtype=np.float64)
    out = np.empty((r.size * s.size, 4), dtype=np.float64)
    call("frcnn_generate_anchors", int(base_size), r.ctypes.data_as(ctypes.c_void_p), r.size,
         s.ctypes.data_as(ctypes.c_void_p), s.size, out.ctypes.data_as(ctypes.c_void_p))
    return out


def generate_anchors_pre(height, width, feat_stride, base_d, out=None):
    """DEVICE f32 [H*W*A,4] (lib/layer_utils/snippets.py:14-30).  base_d: float64 CUDA [A,4]."""
    A = base_d.shape[0]
    _chk(base_d, torch.float64)
    if out is None:
        out = _empty((height * width * A, 4), dtype=torch.float32, device=base_d.device)
    call("frcnn_generate_anchors_pre", height, width, int(feat_stride), _ptr(base_d), A, _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------------------ NMS
def nms(dets, thresh, max_keep=None, keep=None, num=None, rule=NMS_RULE_CPU):
    """cpu_nms (rule NMS_RULE_CPU) / gpu_nms (NMS_RULE_GPU) semantics on device.  dets f32 [k,5] any order ->
    (keep int32 [max_keep], num int32 [1])."""
    _chk(dets)
    k = dets.shape[0]
    max_keep = k if max_keep is None else min(max_keep, k)
    dev = dets.device
    keep = _empty((max(max_keep, 1),), dtype=torch.int32, device=dev) if keep is None else keep
    num = _zeros((1,), dtype=torch.int32, device=dev) if num is None else num
    nb = lib().frcnn_nms_workspace_bytes(max(k, 1))
    ws = workspace(nb, dev, "nms")
    call("frcnn_nms_rule", _ptr(dets), k, float(thresh), int(rule), max_keep, _ptr(keep), _ptr(num), _ptr(ws), ws.numel(), _stream())
    return keep, num


def nms_sorted(boxes, thresh, max_keep=None, rule=NMS_RULE_CPU):
    """Device-pointer form of `_nms`: boxes f32 [k, >=4] sorted by descending score."""
    _chk(boxes)
    k, stride = boxes.shape
    max_keep = k if max_keep is None else min(max_keep, k)
    dev = boxes.device
    keep = _empty((max(max_keep, 1),), dtype=torch.int32, device=dev)
    num = _zeros((1,), dtype=torch.int32, device=dev)
    nb = lib().frcnn_nms_workspace_bytes(max(k, 1))
    ws = workspace(nb, dev, "nms")
    call("frcnn_nms_sorted_rule", _ptr(boxes), k, stride, float(thresh), int(rule), max_keep, _ptr(keep), _ptr(num), _ptr(ws),
         ws.numel(), _stream())
    return keep, num


def non_max_suppression(boxes, scores, max_output_size, iou_threshold, selected=None, num=None):
    """tf.image.non_max_suppression(boxes [k,4], scores [k], max_output_size, iou_threshold) on device (the NMS of the
    reference's USE_E2E_TF graph, proposal_layer.py:72).  -> (selected int32 [max_output_size], num int32 [1])."""
    _chk(boxes), _chk(scores)
    k = boxes.shape[0]
    assert boxes.shape == (k, 4) and scores.numel() == k
    m = min(int(max_output_size), k)
    dev = boxes.device
    selected = _empty((max(m, 1),), dtype=torch.int32, device=dev) if selected is None else selected
    num = _zeros((1,), dtype=torch.int32, device=dev) if num is None else num
    nb = lib().frcnn_nms_workspace_bytes(max(k, 1))
    ws = workspace(nb, dev, "nms")
    call("frcnn_non_max_suppression", _ptr(boxes), _ptr(scores), k, m, float(iou_threshold), _ptr(selected), _ptr(num),
         _ptr(ws), ws.numel(), _stream())
    return selected, num


def bbox_overlaps(boxes, query):
    _chk(boxes, torch.float64), _chk(query, torch.float64)
    out = _empty((boxes.shape[0], query.shape[0]), dtype=torch.float64, device=boxes.device)
    call("frcnn_bbox_overlaps", _ptr(boxes), boxes.shape[0], _ptr(query), query.shape[0], _ptr(out), _stream())
    return out


def bbox_transform_inv(boxes, deltas, out=None):
    """lib/model/bbox_transform.py:35-65 on device: boxes [N,4], deltas [N,4k] -> [N,4k]."""
    _chk(boxes), _chk(deltas)
    N, k = boxes.shape[0], deltas.shape[1] // 4
    out = _empty((N, 4 * k), dtype=torch.float32, device=boxes.device) if out is None else out
    call("frcnn_bbox_transform_inv", _ptr(boxes), _ptr(deltas), N, k, _ptr(out), _stream())
    return out


def clip_boxes(boxes, im_h, im_w):
    """lib/model/bbox_transform.py:68-81 in place: boxes [N,4k]."""
    _chk(boxes)
    call("frcnn_clip_boxes", _ptr(boxes), boxes.shape[0], boxes.shape[1] // 4, float(im_h), float(im_w), _stream())
    return boxes


def bbox_transform(ex_rois, gt_rois):
    """lib/model/bbox_transform.py:14-32 on device: [N,4] x [N,4] -> targets [N,4]."""
    _chk(ex_rois), _chk(gt_rois)
    out = _empty((ex_rois.shape[0], 4), dtype=torch.float32, device=ex_rois.device)
    call("frcnn_bbox_transform", _ptr(ex_rois), _ptr(gt_rois), ex_rois.shape[0], _ptr(out), _stream())
    return out


# ------------------------------------------------------------------------------------------ proposals
def proposal_layer(rpn_cls_prob, rpn_bbox_pred, im_h, im_w, feat_stride, base_d, pre_nms_topn, post_nms_topn,
                   nms_thresh, rois=None, scores=None, num=None, rule=NMS_RULE_CPU):
    """lib/layer_utils/proposal_layer.py:16-53 on device, for the B images of rpn_cls_prob [B,H,W,2A] in one set of launches.
    Returns (rois [B*post,5] with rois[:,0] = image index, scores [B*post,1], num [B])."""
    _chk(rpn_cls_prob), _chk(rpn_bbox_pred), _chk(base_d, torch.float64)
    B, H, W, A2 = rpn_cls_prob.shape
    A = A2 // 2
    dev = rpn_cls_prob.device
    rois = _empty((B * post_nms_topn, 5), dtype=torch.float32, device=dev) if rois is None else rois
    scores = _empty((B * post_nms_topn, 1), dtype=torch.float32, device=dev) if scores is None else scores
    num = _zeros((B,), dtype=torch.int32, device=dev) if num is None else num
    nb = lib().frcnn_proposal_batched_workspace_bytes(B, H, W, A, int(pre_nms_topn))
    ws = workspace(nb, dev, "proposal")
    call("frcnn_proposal_layer_batched", _ptr(rpn_cls_prob), _ptr(rpn_bbox_pred), B, float(im_h), float(im_w), H, W, A,
         int(feat_stride), _ptr(base_d), int(pre_nms_topn), int(post_nms_topn), float(nms_thresh), int(rule), _ptr(rois),
         _ptr(scores), _ptr(num), _ptr(ws), ws.numel(), _stream())
    return rois, scores, num


def proposal_layer_tf(rpn_cls_prob, rpn_bbox_pred, im_h, im_w, feat_stride, base_d, post_nms_topn, nms_thresh, rois=None,
                      scores=None, num=None):
    """lib/layer_utils/proposal_layer.py:56-84 (USE_E2E_TF): NMS with tf.image.non_max_suppression semantics over ALL
    anchors, no pre-NMS top-N, B images per launch.  Returns (rois [B*post,5] zero padded, scores [B*post,1], num [B])."""
    _chk(rpn_cls_prob), _chk(rpn_bbox_pred), _chk(base_d, torch.float64)
    B, H, W, A2 = rpn_cls_prob.shape
    A = A2 // 2
    dev = rpn_cls_prob.device
    rois = _empty((B * post_nms_topn, 5), dtype=torch.float32, device=dev) if rois is None else rois
    scores = _empty((B * post_nms_topn, 1), dtype=torch.float32, device=dev) if scores is None else scores
    num = _zeros((B,), dtype=torch.int32, device=dev) if num is None else num
    nb = lib().frcnn_proposal_batched_workspace_bytes(B, H, W, A, 0)
    ws = workspace(nb, dev, "proposal")
    call("frcnn_proposal_layer_tf_batched", _ptr(rpn_cls_prob), _ptr(rpn_bbox_pred), B, float(im_h), float(im_w), H, W, A,
         int(feat_stride), _ptr(base_d), int(post_nms_topn), float(nms_thresh), _ptr(rois), _ptr(scores), _ptr(num),
         _ptr(ws), ws.numel(), _stream())
    return rois, scores, num


def proposal_top_layer_inds(rpn_cls_prob, rpn_bbox_pred, im_h, im_w, feat_stride, base_d, top_inds, rois=None, scores=None):
    """proposal_top_layer.py:30-33,46-55 for caller-chosen anchor indices (int32 device tensor): decode + clip those."""
    _chk(rpn_cls_prob), _chk(rpn_bbox_pred), _chk(base_d, torch.float64), _chk(top_inds, torch.int32)
    _, H, W, A2 = rpn_cls_prob.shape
    n = top_inds.numel()
    dev = rpn_cls_prob.device
    rois = _empty((n, 5), dtype=torch.float32, device=dev) if rois is None else rois
    scores = _empty((n, 1), dtype=torch.float32, device=dev) if scores is None else scores
    call("frcnn_proposal_top_layer_inds", _ptr(rpn_cls_prob), _ptr(rpn_bbox_pred), float(im_h), float(im_w), H, W, A2 // 2,
         int(feat_stride), _ptr(base_d), _ptr(top_inds), n, _ptr(rois), _ptr(scores), _stream())
    return rois, scores


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, im_h, im_w, feat_stride, base_d, rpn_top_n, rois=None, scores=None):
    _chk(rpn_cls_prob), _chk(rpn_bbox_pred), _chk(base_d, torch.float64)
    _, H, W, A2 = rpn_cls_prob.shape
    A = A2 // 2
    dev = rpn_cls_prob.device
    rois = _empty((rpn_top_n, 5), dtype=torch.float32, device=dev) if rois is None else rois
    scores = _empty((rpn_top_n, 1), dtype=torch.float32, device=dev) if scores is None else scores
    nb = lib().frcnn_proposal_workspace_bytes(H, W, A, int(rpn_top_n))
    ws = workspace(nb, dev, "proposal")
    call("frcnn_proposal_top_layer", _ptr(rpn_cls_prob), _ptr(rpn_bbox_pred), float(im_h), float(im_w), H, W, A,
         int(feat_stride), _ptr(base_d), int(rpn_top_n), _ptr(rois), _ptr(scores), _ptr(ws), ws.numel(), _stream())
    return rois, scores


def crop_and_resize(feat, rois, feat_stride, pool, max_pool=False, out=None):
    """feat [N,H,W,C] or [H,W,C]; rois [R,5] with rois[:,0] = image index (the box_ind of tf.image.crop_and_resize,
    network.py:143) -> [R,pool,pool,C] (TF crop_and_resize semantics)."""
    _chk(feat), _chk(rois)
    H, W, C = feat.shape[-3:]
    N = feat.shape[0] if feat.dim() == 4 else 1
    R = rois.shape[0]
    out = _empty((R, pool, pool, C), dtype=torch.float32, device=feat.device) if out is None else out
    call("frcnn_crop_and_resize_batched", _ptr(feat), N, H, W, C, _ptr(rois), R, float(feat_stride), int(pool),
         1 if max_pool else 0, _ptr(out), _stream())
    return out


def crop_and_resize_bias_act(feat, rois, feat_stride, pool, bias, act, out=None):
    """act(crop_and_resize(feat) + bias): see frcnn_crop_and_resize_bias_act."""
    _chk(feat), _chk(rois)
    H, W, C = feat.shape[-3:]
    N = feat.shape[0] if feat.dim() == 4 else 1
    R = rois.shape[0]
    out = _empty((R, pool, pool, C), dtype=torch.float32, device=feat.device) if out is None else out
    call("frcnn_crop_and_resize_bias_act", _ptr(feat), N, H, W, C, _ptr(rois), R, float(feat_stride), int(pool), _ptr(bias),
         int(act), _ptr(out), _stream())
    return out


def detect_post(cls_prob, bbox_pred, rois, num_rois, im_scale, im_h, im_w, nms_thresh=0.3, score_thresh=0.0,
                max_per_image=100, max_out=None, out=None, count=None, batch=1, rule=NMS_RULE_CPU):
    """lib/model/test.py:95-102 + :162-180 on device.  batch = 1: (dets [max_out,6], count [1]).  batch = B: cls_prob
    [B*R,C] (R rows per image), num_rois [B] -> (dets [B,max_out,6], count [B]), one launch pair for all images.
    bbox_pred = None: cfg.TEST.BBOX_REG False (the rois themselves, un-regressed and un-clipped, test.py:103-105)."""
    _chk(cls_prob), _chk(rois)
    if bbox_pred is not None:
        _chk(bbox_pred)
    B = int(batch)
    R, C = cls_prob.shape[0] // B, cls_prob.shape[1]
    if cls_prob.shape[0] != B * R or rois.shape[0] != B * R:
        raise ValueError("detect_post: %d rows do not split into %d images" % (cls_prob.shape[0], B))
    dev = cls_prob.device
    max_out = (max_per_image + 28 if max_per_image > 0 else R * (C - 1)) if max_out is None else max_out
    shape = (max_out, 6) if B == 1 else (B, max_out, 6)
    out = _empty(shape, dtype=torch.float32, device=dev) if out is None else out
    count = _zeros((B,), dtype=torch.int32, device=dev) if count is None else count
    # a batched record may be a strided view (per-image slices contiguous, frcnn_hip.parallel.new_record)
    if out.dtype != torch.float32 or out.shape[-1] != 6 or out.stride(-1) != 1 or out.stride(-2) != 6 or out.shape[-2] < max_out:
        raise ValueError("detect_post: `out` must be float32 [.., >=max_out, 6] with contiguous images")
    out_stride = out.stride(0) if out.dim() == 3 else 0
    nb = lib().frcnn_detect_post_batched_workspace_bytes(B, R, C)
    ws = workspace(nb, dev, "detect_post")
    call("frcnn_detect_post_batched", _ptr(cls_prob), _ptr(bbox_pred), _ptr(rois), _ptr(num_rois), B, R, C, float(im_scale),
         int(im_h), int(im_w), float(nms_thresh), int(rule), float(score_thresh), int(max_per_image), _ptr(out), _ptr(count),
         int(max_out), int(out_stride), _ptr(ws), ws.numel(), _stream())
    return out, count


def im_detect_boxes(rois, bbox_pred, im_scale, im_h, im_w, num_classes=None):
    """lib/model/test.py:95-105 on device: -> pred_boxes [R,4C].  bbox_pred = None (cfg.TEST.BBOX_REG False): np.tile(boxes, (1, C))."""
    _chk(rois)
    if bbox_pred is None:
        R, C4 = rois.shape[0], 4 * int(num_classes)
    else:
        _chk(bbox_pred)
        R, C4 = bbox_pred.shape
    out = _empty((R, C4), dtype=torch.float32, device=rois.device)
    call("frcnn_im_detect_boxes", _ptr(rois), _ptr(bbox_pred), R, C4 // 4, float(im_scale), int(im_h), int(im_w), _ptr(out),
         _stream())
    return out


# ------------------------------------------------------------------------------------------ dense
def pack_filter_hwio(w_hwio, scale=None):
    """HOST numpy HWIO -> packed [Cout][KH][KW][Cin] (folding a per-output-channel scale)."""
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    KH, KW, Cin, Cout = w.shape
    out = np.empty((Cout, KH, KW, Cin), dtype=np.float32)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    call("frcnn_pack_filter_hwio", w.ctypes.data_as(ctypes.c_void_p), KH, KW, Cin, Cout,
         None if sc is None else sc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def pack_filter_foldw(w_hwio, scale=None):
    """HOST: stem filter HWIO [KH,KW<=8,Cin<=4,Cout] -> [Cout][KH][8][4] (zero padded) for the
    fold_w mode of frcnn_conv2d_nhwc: the kw taps become part of the contiguous channel run of the
    4-channel-padded image, so the 7x7x3 stem runs on the MFMA pipe with K = KH*32."""
    w = np.asarray(w_hwio, dtype=np.float32)
    KH, KW, Cin, Cout = w.shape
    assert KW <= 8 and Cin <= 4
    if scale is not None:
        w = w * np.asarray(scale, dtype=np.float32)[None, None, None, :]
    out = np.zeros((Cout, KH, 8, 4), dtype=np.float32)
    out[:, :, :KW, :Cin] = np.transpose(w, (3, 0, 1, 2))
    return out


def conv_out_size(n, k, stride, pad_lo, pad_hi):
    return (n + pad_lo + pad_hi - k) // stride + 1


split_k = True      # module switch: allow split-K launches (frcnn_conv2d_nhwc_ws) where the library's plan asks for them


def conv2d(x, w_packed, bias, KH, KW, stride=1, pad=(0, 0, 0, 0), act=ACT_NONE, residual=None, res_stride=1,
           fold_w=False, out=None, mask=None):
    """x [N,H,W,Cin]; w_packed [Cout,KH,KW,Cin] (fold_w: [Cout,KH,8,4]); pad = (top, bottom, left, right).
    mask (training, a float32 tensor of the result's shape): out = mask > 0 ? out : 0 inside the launch (frcnn_conv2d_nhwc_masked_ws)."""
    _chk(x), _chk(w_packed)
    N, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    OH = conv_out_size(H, KH, stride, pad[0], pad[1])
    OW = conv_out_size(W, KW, stride, pad[2], pad[3])
    out = _empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device) if out is None else out
    RH = RW = 0
    if residual is not None:
        _chk(residual)
        RH, RW = residual.shape[1], residual.shape[2]
    nb = lib().frcnn_conv2d_workspace_bytes(N, OH, OW, Cout, KH, KW, Cin, 1 if fold_w else 0) if split_k else 0
    if mask is not None:
        _chk(mask)
        assert not fold_w and mask.numel() == out.numel()
        ws = workspace(nb, x.device, "conv_splitk") if nb else None
        call("frcnn_conv2d_nhwc_masked_ws", _ptr(x), N, H, W, Cin, _ptr(w_packed), _ptr(bias), _ptr(residual), RH, RW,
             int(res_stride), _ptr(out), OH, OW, Cout, KH, KW, int(stride), int(pad[0]), int(pad[2]), int(act), _ptr(mask),
             _ptr(ws), ws.numel() if nb else 0, _stream())
        return out
    if nb:                                  # under-filled launch: split-K through a per-chain scratch buffer
        ws = workspace(nb, x.device, "conv_splitk")
        call("frcnn_conv2d_nhwc_ws", _ptr(x), N, H, W, Cin, _ptr(w_packed), _ptr(bias), _ptr(residual), RH, RW,
             int(res_stride), _ptr(out), OH, OW, Cout, KH, KW, int(stride), int(pad[0]), int(pad[2]), int(act),
             1 if fold_w else 0, _ptr(ws), ws.numel(), _stream())
        return out
    call("frcnn_conv2d_nhwc", _ptr(x), N, H, W, Cin, _ptr(w_packed), _ptr(bias), _ptr(residual), RH, RW,
         int(res_stride), _ptr(out), OH, OW, Cout, KH, KW, int(stride), int(pad[0]), int(pad[2]), int(act),
         1 if fold_w else 0, _stream())
    return out


def prep_image_shape(h, w, target_size, max_size):
    """HOST: (im_scale, OH, OW) of _get_image_blob / prep_im_for_blob for an h x w image."""
    sc, oh, ow = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
    call("frcnn_prep_image_shape", int(h), int(w), int(target_size), int(max_size), ctypes.byref(sc), ctypes.byref(oh),
         ctypes.byref(ow))
    return sc.value, oh.value, ow.value


def prep_image(im_d, pixel_means, im_scale, out_hw, out=None, out_c=4):
    """im_d: BGR uint8 or float32 [h,w,3] on device -> float32 [1,OH,OW,out_c] = resize(im - PIXEL_MEANS) (cv2 INTER_LINEAR)."""
    assert im_d.is_cuda and im_d.is_contiguous() and im_d.dim() == 3 and im_d.shape[2] == 3
    assert im_d.dtype in (torch.uint8, torch.float32)
    h, w = im_d.shape[:2]
    OH, OW = out_hw
    out = _empty((1, OH, OW, out_c), dtype=torch.float32, device=im_d.device) if out is None else out
    assert out.shape == (1, OH, OW, out_c)
    means = (ctypes.c_double * 3)(*[float(v) for v in np.asarray(pixel_means, dtype=np.float64).reshape(-1)[:3]])
    call("frcnn_prep_image", _ptr(im_d), 1 if im_d.dtype == torch.float32 else 0, h, w, means, float(im_scale), _ptr(out), OH, OW,
         int(out_c), _stream())
    return out


def winograd_filter_transform(w_hwio, scale=None, m=2):
    """HOST: 3x3 HWIO filter -> U [(m+2)^2, Cout, Cin] (F(m x m,3x3), optional folded per-output scale)."""
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    assert w.shape[0] == 3 and w.shape[1] == 3
    Cin, Cout = w.shape[2], w.shape[3]
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    if m == 7:                               # the mixed F(4,3)+F(3,3) scheme for 7x7 maps: 121 points
        out = np.empty((121, Cout, Cin), dtype=np.float32)
        call("frcnn_winograd7_filter_transform", w.ctypes.data_as(ctypes.c_void_p), Cin, Cout,
             None if sc is None else sc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        return out
    out = np.empty(((m + 2) ** 2, Cout, Cin), dtype=np.float32)
    call("frcnn_winograd_filter_transform", w.ctypes.data_as(ctypes.c_void_p), Cin, Cout,
         None if sc is None else sc.ctypes.data_as(ctypes.c_void_p), int(m), out.ctypes.data_as(ctypes.c_void_p))
    return out


def winograd_filter_transform_device(w_packed, m=4, transpose_flip=False, out=None):
    """Packed filter [Cout,3,3,Cin] on device -> U [(m+2)^2,Cout,Cin] (or, transpose_flip, the dgrad filter [(m+2)^2,Cin,Cout])."""
    _chk(w_packed)
    O, kh, kw, C = w_packed.shape
    assert kh == 3 and kw == 3
    G = winograd_points(m)
    shape = (G, C, O) if transpose_flip else (G, O, C)
    out = _empty(shape, dtype=torch.float32, device=w_packed.device) if out is None else out
    assert out.numel() == G * O * C
    if m == 7:
        call("frcnn_winograd7_filter_transform_device", _ptr(w_packed), O, C, 1 if transpose_flip else 0, _ptr(out), _stream())
    else:
        call("frcnn_winograd_filter_transform_device", _ptr(w_packed), O, C, int(m), 1 if transpose_flip else 0, _ptr(out), _stream())
    return out.view(shape)


def winograd_points(m):
    """GEMMs per layer: (m+2)^2 for F(m x m,3x3); 121 for the mixed 7x7 scheme (m == 7)."""
    return 121 if m == 7 else (m + 2) ** 2


def winograd_tiles(N, H, W, m):
    if m == 7:
        assert H == 7 and W == 7
        return N
    return N * ((H + m - 1) // m) * ((W + m - 1) // m)


def winograd_input_transform(x, v, m=2):
    _chk(x), _chk(v)
    N, H, W, C = x.shape
    assert v.numel() == winograd_points(m) * winograd_tiles(N, H, W, m) * C
    if m == 7:
        call("frcnn_winograd7_input_transform", _ptr(x), N, C, _ptr(v), _stream())
    else:
        call("frcnn_winograd_input_transform", _ptr(x), N, H, W, C, int(m), _ptr(v), _stream())
    return v


def gemm_batched_nt(x, w, y):
    """y[g] = x[g] @ w[g]^T for x [G,M,K], w [G,N,K], y [G,M,N] in one launch of the f32-MFMA kernel."""
    _chk(x), _chk(w), _chk(y)
    G, M, K = x.shape
    N = w.shape[1]
    assert w.shape == (G, N, K) and y.shape == (G, M, N)
    call("frcnn_gemm_batched_nt", _ptr(x), _ptr(w), _ptr(y), G, M, N, K, _stream())
    return y


def winograd_output_transform(mm, bias, act, out, m=2):
    _chk(mm), _chk(out)
    N, H, W, C = out.shape
    assert mm.numel() == winograd_points(m) * winograd_tiles(N, H, W, m) * C
    if m == 7:
        call("frcnn_winograd7_output_transform", _ptr(mm), N, C, _ptr(bias), int(act), _ptr(out), _stream())
    else:
        call("frcnn_winograd_output_transform", _ptr(mm), N, H, W, C, int(m), _ptr(bias), int(act), _ptr(out), _stream())
    return out


def winograd_output_transform_masked(mm, mask, out, m, out_planes=None):
    """Training: out = mask > 0 ? A^T M A : 0 (no bias, no activation) as float32 `out` [N,H,W,C] and, with `out_planes` (an H2 of
    [N*H*W, C]), as the operand planes of the next data-gradient GEMM too.  m in (4, 7)."""
    _chk(mm), _chk(mask), _chk(out)
    N, H, W, C = out.shape
    assert mm.numel() == winograd_points(m) * winograd_tiles(N, H, W, m) * C and mask.numel() == out.numel()
    if out_planes is not None:
        assert isinstance(out_planes, H2) and out_planes.rows == N * H * W and out_planes.K == C
    pl, inv = (None, None) if out_planes is None else (out_planes.planes, out_planes.inv)
    if m == 7:
        call("frcnn_winograd7_output_transform_masked", _ptr(mm), N, C, _ptr(mask), _ptr(out), _ptr(pl), _ptr(inv), _stream())
    else:
        call("frcnn_winograd_output_transform_masked", _ptr(mm), N, H, W, C, int(m), _ptr(mask), _ptr(out), _ptr(pl), _ptr(inv), _stream())
    return out


def conv3x3_winograd(x, u, bias, act=ACT_NONE, out=None, v_buf=None, m_buf=None, u_planes=None, v_planes=None, mask=None, out_planes=None):
    """3x3 / stride 1 / pad 1 convolution as Winograd F(m x m,3x3): x [N,H,W,Cin], u [(m+2)^2,Cout,Cin] -> [N,H,W,Cout].
    u_planes = h2_pack_w(u) and v_planes (an H2 of [points * tiles, Cin]): the input transform emits V as operand planes and the
    products run in frcnn_gemm_h2 instead of the f32-MFMA batched GEMM.
    mask / out_planes (training, m in (4, 7), no bias / activation): winograd_output_transform_masked."""
    N, H, W, Cin = x.shape
    G, Cout = u.shape[0], u.shape[1]
    m = {16: 2, 36: 4, 121: 7}[G]
    T = winograd_tiles(N, H, W, m)
    dev = x.device
    mm = _empty((G, T, Cout), dtype=torch.float32, device=dev) if m_buf is None else m_buf
    out = _empty((N, H, W, Cout), dtype=torch.float32, device=dev) if out is None else out
    if u_planes is not None:
        v_planes = H2.empty(G * T, Cin, dev) if v_planes is None else v_planes
        winograd_input_transform_h2(x, v_planes, m)
        gemm_h2(v_planes, u_planes, G, T, Cout, Cin, out=mm.view(G * T, Cout))
    else:
        v = _empty((G, T, Cin), dtype=torch.float32, device=dev) if v_buf is None else v_buf
        winograd_input_transform(x, v, m)
        gemm_batched_nt(v, u, mm)
    if mask is not None:
        assert bias is None and act == ACT_NONE
        return winograd_output_transform_masked(mm, mask, out, m, out_planes)
    if out_planes is not None:              # float32 result + the operand planes of the convolution that follows (training forward)
        winograd_output_transform_h2(mm, bias, act, tuple(out.shape), m, out_planes, out)
        return out
    return winograd_output_transform(mm, bias, act, out, m)


def maxpool(x, k, stride, pad=(0, 0, 0, 0), out=None):
    _chk(x)
    N, H, W, C = x.shape
    OH = conv_out_size(H, k, stride, pad[0], pad[1])
    OW = conv_out_size(W, k, stride, pad[2], pad[3])
    out = _empty((N, OH, OW, C), dtype=torch.float32, device=x.device) if out is None else out
    call("frcnn_maxpool_nhwc", _ptr(x), N, H, W, C, int(k), int(stride), int(pad[0]), int(pad[2]), _ptr(out), OH, OW,
         _stream())
    return out


def dwconv3x3(x, w, bias, stride=1, pad=(1, 1, 1, 1), act=ACT_NONE, out=None, out_planes=None, want_f32=True):
    """Depthwise 3x3 (+ folded BN bias, activation).  out_planes: an H2 of [N*OH*OW, C] that receives the result as frcnn_gemm_h2
    operand planes (frcnn_dwconv3x3_nhwc_h2); want_f32 = False then skips the float32 tensor (returns None for it)."""
    _chk(x), _chk(w)
    N, H, W, C = x.shape
    OH = conv_out_size(H, 3, stride, pad[0], pad[1])
    OW = conv_out_size(W, 3, stride, pad[2], pad[3])
    if out_planes is None:
        out = _empty((N, OH, OW, C), dtype=torch.float32, device=x.device) if out is None else out
        call("frcnn_dwconv3x3_nhwc", _ptr(x), N, H, W, C, _ptr(w), _ptr(bias), _ptr(out), OH, OW, int(stride),
             int(pad[0]), int(pad[2]), int(act), _stream())
        return out
    assert isinstance(out_planes, H2) and out_planes.rows == N * OH * OW and out_planes.K == C
    if want_f32 and out is None:
        out = _empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    call("frcnn_dwconv3x3_nhwc_h2", _ptr(x), N, H, W, C, _ptr(w), _ptr(bias), _ptr(out if want_f32 else None), _ptr(out_planes.planes),
         _ptr(out_planes.inv), OH, OW, int(stride), int(pad[0]), int(pad[2]), int(act), _stream())
    return out if want_f32 else None


def spatial_mean(x, out=None):
    _chk(x)
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    out = _empty((N, C), dtype=torch.float32, device=x.device) if out is None else out
    call("frcnn_spatial_mean", _ptr(x), N, HW, C, _ptr(out), _stream())
    return out


def softmax_rows(x, C=None, out=None):
    _chk(x)
    R, ld = x.shape
    C = ld if C is None else C
    out = _empty((R, C), dtype=torch.float32, device=x.device) if out is None else out
    call("frcnn_softmax_rows", _ptr(x), R, C, ld, _ptr(out), _stream())
    return out


def rpn_softmax(score, A, out=None):
    """score [1,H,W,ld] (first 2A channels = bg|fg scores) -> prob [1,H,W,2A]."""
    _chk(score)
    N, H, W, ld = score.shape
    out = _empty((N, H, W, 2 * A), dtype=torch.float32, device=score.device) if out is None else out
    call("frcnn_rpn_softmax", _ptr(score), N * H * W, A, ld, _ptr(out), _stream())
    return out


def copy_cols(src, col0, cols, out=None):
    _chk(src)
    ld = src.shape[-1]
    R = src.numel() // ld
    out = _empty(tuple(src.shape[:-1]) + (cols,), dtype=torch.float32, device=src.device) if out is None else out
    call("frcnn_copy_cols", _ptr(src), R, ld, int(col0), int(cols), _ptr(out), cols, _stream())
    return out


# ------------------------------------------------------------------------------------------ training
def _host_doubles(values):
    a = np.ascontiguousarray(values, dtype=np.float64)
    if _binding.recorder is not None:
        _binding.recorder.keep.append(a)         # (a recorded launch keeps the pointer)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def rpn_target_opts(clobber_positives=False, positive_weight=-1.0, inside_weights=(1.0, 1.0, 1.0, 1.0)):
    """opts block of frcnn_anchor_target_layer*: TRAIN.RPN_CLOBBER_POSITIVES, TRAIN.RPN_POSITIVE_WEIGHT, TRAIN.RPN_BBOX_INSIDE_WEIGHTS."""
    return [1.0 if clobber_positives else 0.0, float(positive_weight)] + [float(v) for v in inside_weights]


def roi_target_opts(use_gt=False, inside_weights=(1.0, 1.0, 1.0, 1.0)):
    """opts block of frcnn_proposal_target_layer*: TRAIN.USE_GT, TRAIN.BBOX_INSIDE_WEIGHTS."""
    return [1.0 if use_gt else 0.0] + [float(v) for v in inside_weights]


def anchor_target_layer(gt_boxes, im_h, im_w, H, W, base_d, feat_stride=16, rpn_batchsize=256, fg_fraction=0.5,
                        pos_overlap=0.7, neg_overlap=0.3, seed=0, opts=None):
    """lib/layer_utils/anchor_target_layer.py:18-138 on device -> (labels [1,1,A*H,W], targets, inside, outside [1,H,W,4A]).
    opts: rpn_target_opts(...) or None (reference defaults)."""
    _chk(gt_boxes), _chk(base_d, torch.float64)
    A, G, dev = base_d.shape[0], gt_boxes.shape[0], gt_boxes.device
    labels = _empty((1, 1, A * H, W), dtype=torch.float32, device=dev)
    tg, iw, ow = (_empty((1, H, W, 4 * A), dtype=torch.float32, device=dev) for _ in range(3))
    # (sized for every box count the tensor's storage can hold: a recorded step is replayed with the next image's G)
    Gcap = max(G, (gt_boxes.untyped_storage().nbytes() - 4 * gt_boxes.storage_offset()) // 20)
    ws = workspace(lib().frcnn_anchor_target_workspace_bytes(H, W, A, Gcap), dev, "anchor_target")
    o = _host_doubles(opts) if opts is not None else (None, None)
    call("frcnn_anchor_target_layer", _ptr(gt_boxes), G, float(im_h), float(im_w), H, W, A, int(feat_stride), _ptr(base_d),
         int(rpn_batchsize), float(fg_fraction), float(pos_overlap), float(neg_overlap), int(seed), o[1], _ptr(labels), _ptr(tg),
         _ptr(iw), _ptr(ow), _ptr(ws), ws.numel(), _stream())
    if _binding.recorder is not None:
        _binding.recorder.patch_last(1, var="gt")    # the image's box count (gt_boxes: the first G rows of a static buffer)
        if int(seed) >= 0:
            _binding.recorder.patch_last(13)         # the sampling seed advances with the step (Network._sample_seed)
    return labels, tg, iw, ow


def anchor_target_layer_inject(gt_boxes, im_h, im_w, H, W, base_d, disable, feat_stride=16, rpn_batchsize=256, fg_fraction=0.5,
                               pos_overlap=0.7, neg_overlap=0.3, opts=None):
    """anchor_target_layer with the reference's host-drawn `disable_inds` (int32 device tensor of ALL-anchor indices, may be
    empty): see frcnn_anchor_target_layer_inject."""
    _chk(gt_boxes), _chk(base_d, torch.float64), _chk(disable, torch.int32)
    A, G, dev = base_d.shape[0], gt_boxes.shape[0], gt_boxes.device
    labels = _empty((1, 1, A * H, W), dtype=torch.float32, device=dev)
    tg, iw, ow = (_empty((1, H, W, 4 * A), dtype=torch.float32, device=dev) for _ in range(3))
    ws = workspace(lib().frcnn_anchor_target_workspace_bytes(H, W, A, G), dev, "anchor_target")
    o = _host_doubles(opts) if opts is not None else (None, None)
    call("frcnn_anchor_target_layer_inject", _ptr(gt_boxes), G, float(im_h), float(im_w), H, W, A, int(feat_stride), _ptr(base_d),
         int(rpn_batchsize), float(fg_fraction), float(pos_overlap), float(neg_overlap), _ptr(disable) if disable.numel() else None,
         int(disable.numel()), o[1], _ptr(labels), _ptr(tg), _ptr(iw), _ptr(ow), _ptr(ws), ws.numel(), _stream())
    return labels, tg, iw, ow


def proposal_target_layer_inject(rpn_rois, rpn_scores, gt_boxes, num_classes, keep_inds, n_fg, means=(0.0, 0.0, 0.0, 0.0),
                                 stds=(0.1, 0.1, 0.2, 0.2), opts=None):
    """proposal_target_layer for host-drawn keep_inds (int32 device [batch], fg rows first): frcnn_proposal_target_layer_inject.
    opts: roi_target_opts(...) or None."""
    _chk(rpn_rois), _chk(rpn_scores), _chk(gt_boxes), _chk(keep_inds, torch.int32)
    dev, N, G, B, C = rpn_rois.device, rpn_rois.shape[0], gt_boxes.shape[0], keep_inds.numel(), int(num_classes)
    rois = _empty((B, 5), dtype=torch.float32, device=dev)
    sc = _empty((B,), dtype=torch.float32, device=dev)
    labels = _empty((B, 1), dtype=torch.float32, device=dev)
    tg, iw, ow = (_empty((B, 4 * C), dtype=torch.float32, device=dev) for _ in range(3))
    m, s, o = _host_doubles(means), _host_doubles(stds), (_host_doubles(opts) if opts is not None else (None, None))
    call("frcnn_proposal_target_layer_inject", _ptr(rpn_rois), _ptr(rpn_scores), N, _ptr(gt_boxes), G, C, B, _ptr(keep_inds), int(n_fg),
         m[1], s[1], o[1], _ptr(rois), _ptr(sc), _ptr(labels), _ptr(tg), _ptr(iw), _ptr(ow), _stream())
    return rois, sc, labels, tg, iw, ow


def proposal_target_layer(rpn_rois, rpn_scores, gt_boxes, num_classes, batch_size=256, fg_fraction=0.25, fg_thresh=0.5,
                          bg_hi=0.5, bg_lo=0.0, means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2), seed=0, num=None, opts=None):
    """lib/layer_utils/proposal_target_layer.py:18-152 on device.  num: int32 [1] device tensor = valid rows of rpn_rois
    (the proposal layer's count); without it every row is a proposal.  opts: roi_target_opts(...) or None."""
    _chk(rpn_rois), _chk(rpn_scores), _chk(gt_boxes)
    dev, N, G, B, C = rpn_rois.device, rpn_rois.shape[0], gt_boxes.shape[0], int(batch_size), int(num_classes)
    rois = _empty((B, 5), dtype=torch.float32, device=dev)
    sc = _empty((B,), dtype=torch.float32, device=dev)
    labels = _empty((B, 1), dtype=torch.float32, device=dev)
    tg, iw, ow = (_empty((B, 4 * C), dtype=torch.float32, device=dev) for _ in range(3))
    counts = _zeros((4,), dtype=torch.int32, device=dev)
    m, s, o = _host_doubles(means), _host_doubles(stds), (_host_doubles(opts) if opts is not None else (None, None))
    if num is not None:
        call("frcnn_proposal_target_layer_dn", _ptr(rpn_rois), _ptr(rpn_scores), N, _ptr(num), _ptr(gt_boxes), G, C, B,
             float(fg_fraction), float(fg_thresh), float(bg_hi), float(bg_lo), m[1], s[1], int(seed), o[1], _ptr(rois), _ptr(sc),
             _ptr(labels), _ptr(tg), _ptr(iw), _ptr(ow), _ptr(counts), _stream())
        if _binding.recorder is not None:
            _binding.recorder.patch_last(5, var="gt")
            _binding.recorder.patch_last(14)
        return rois, sc, labels, tg, iw, ow, counts
    call("frcnn_proposal_target_layer", _ptr(rpn_rois), _ptr(rpn_scores), N, _ptr(gt_boxes), G, C, B, float(fg_fraction),
         float(fg_thresh), float(bg_hi), float(bg_lo), m[1], s[1], int(seed), o[1], _ptr(rois), _ptr(sc), _ptr(labels), _ptr(tg),
         _ptr(iw), _ptr(ow), _ptr(counts), _stream())
    if _binding.recorder is not None:
        _binding.recorder.patch_last(4, var="gt")
        _binding.recorder.patch_last(13)
    return rois, sc, labels, tg, iw, ow, counts


def softmax_ce_loss(logits, labels, rpn_shape=None):
    """-> (loss [1], dlogits like logits).  rpn_shape=(A,H,W): logits [1,H,W,2A] with labels [1,1,A*H,W]."""
    _chk(logits), _chk(labels)
    dev = logits.device
    if rpn_shape is None:
        R, C, A, H, W = logits.shape[0], logits.shape[1], 0, 0, 0
    else:
        A, H, W = rpn_shape
        R, C = A * H * W, 2
    loss = _zeros((1,), dtype=torch.float32, device=dev)
    grad = _zeros(tuple(logits.shape), dtype=logits.dtype, device=logits.device)
    ws = workspace(lib().frcnn_loss_workspace_bytes(R), dev, "loss")
    call("frcnn_softmax_ce_loss", _ptr(logits), _ptr(labels), R, C, A, H, W, _ptr(loss), _ptr(grad), _ptr(ws), ws.numel(), _stream())
    return loss, grad


def smooth_l1_loss(pred, targets, inside_w, outside_w, sigma, mean_divisor):
    _chk(pred), _chk(targets), _chk(inside_w), _chk(outside_w)
    dev, n = pred.device, pred.numel()
    loss = _zeros((1,), dtype=torch.float32, device=dev)
    grad = _empty(tuple(pred.shape), dtype=pred.dtype, device=pred.device)
    ws = workspace(lib().frcnn_loss_workspace_bytes(n), dev, "loss")
    call("frcnn_smooth_l1_loss", _ptr(pred), _ptr(targets), _ptr(inside_w), _ptr(outside_w), n, float(sigma), float(mean_divisor),
         _ptr(loss), _ptr(grad), _ptr(ws), ws.numel(), _stream())
    return loss, grad


# ------------------------------------------------------------------------------------------ backward
def transpose_pad(x2d, Mp, out=None):
    """x2d [M,C] -> [C,Mp] (zero padded)."""
    _chk(x2d)
    M, C = x2d.shape
    out = _empty((C, Mp), dtype=torch.float32, device=x2d.device) if out is None else out
    call("frcnn_transpose_pad", _ptr(x2d), M, C, _ptr(out), int(Mp), _stream())
    return out


def im2col_t(x, KH, KW, stride, pad, OH, OW, Mp, out=None):
    _chk(x)
    N, H, W, Cin = x.shape
    out = _empty((KH * KW * Cin, Mp), dtype=torch.float32, device=x.device) if out is None else out
    call("frcnn_im2col_t", _ptr(x), N, H, W, Cin, OH, OW, KH, KW, int(stride), int(pad[0]), int(pad[2]), _ptr(out), int(Mp), _stream())
    return out


def flip_transpose_filter(w_packed, out=None):
    _chk(w_packed)
    Cout, KH, KW, Cin = w_packed.shape
    out = _empty((Cin, KH, KW, Cout), dtype=torch.float32, device=w_packed.device) if out is None else out
    call("frcnn_flip_transpose_filter", _ptr(w_packed), Cout, KH, KW, Cin, _ptr(out), _stream())
    return out


def conv2d_dgrad_strided(dy, w_packed, stride, pad, H, W, dx, accumulate):
    _chk(dy), _chk(w_packed), _chk(dx)
    N, OH, OW, Cout = dy.shape
    _, KH, KW, Cin = w_packed.shape
    call("frcnn_conv2d_dgrad_strided", _ptr(dy), N, OH, OW, Cout, _ptr(w_packed), KH, KW, Cin, int(stride), int(pad[0]), int(pad[2]),
         _ptr(dx), int(H), int(W), 1 if accumulate else 0, _stream())
    return dx


def conv2d_wgrad_supported(Cin, Cout):
    return bool(lib().frcnn_conv2d_wgrad_supported(int(Cin), int(Cout)))


def conv2d_wgrad(dy, x, KH, KW, stride, pad, out, h2=False):
    """dW [Cout, KH, KW, Cin] of a convolution from dY [N,OH,OW,Cout] and its input X [N,H,W,Cin], both read as they lie; pad =
    (top, bottom, left, right).  Cin % 64 == 0 and Cout % 64 == 0 (conv2d_wgrad_supported).  h2 False: f32 matrix pipe
    (csrc/wgrad_tn.hip); True: two-piece fp16 operands split in registers, fp16 matrix pipe (csrc/wgrad_h2.hip)."""
    _chk(dy), _chk(x), _chk(out)
    N, OH, OW, Cout = dy.shape
    _, H, W, Cin = x.shape
    if out.numel() != Cout * KH * KW * Cin:
        raise ValueError("conv2d_wgrad: out has %d elements, the filter %d" % (out.numel(), Cout * KH * KW * Cin))
    name = "frcnn_conv2d_wgrad_h2" if h2 else "frcnn_conv2d_wgrad"
    nb = getattr(lib(), name + "_workspace_bytes")(N, OH, OW, Cin, Cout, int(KH), int(KW))
    ws = workspace(nb, x.device, "conv_wgrad") if nb else None
    call(name, _ptr(dy), _ptr(x), N, H, W, Cin, OH, OW, Cout, int(KH), int(KW), int(stride), int(pad[0]), int(pad[2]),
         _ptr(out), _ptr(ws), int(nb), _stream())
    return out


def relu_bwd(grad, y):
    _chk(grad), _chk(y)
    call("frcnn_relu_bwd", _ptr(grad), _ptr(y), grad.numel(), _stream())
    return grad


def gemm_x3_pack(w, planes=None):
    """bf16 planes [G][3][N][K] of a filter bank w [G,N,K] or [N,...K] (frcnn_gemm_x3_pack); `planes`: re-split into an existing buffer."""
    _chk(w)
    G, N, K = (w.shape[0], w.shape[1], w.shape[2]) if w.dim() == 3 else (1, w.shape[0], w[0].numel())
    if planes is None:
        planes = torch.empty(lib().frcnn_gemm_x3_pack_bytes(G, N, K), dtype=torch.uint8, device=w.device)
    call("frcnn_gemm_x3_pack", _ptr(w), G, N, K, _ptr(planes), _stream())
    return planes


def gemm_x3(x, planes, G, M, N, K, bias=None, residual=None, act=ACT_NONE, out=None, cfg=-1, terms=6):
    """out[g] = act(x[g] W[g]^T + bias + residual[g]) through frcnn_gemm_x3 (products on the bf16 pipe as exact 3-way splits).
    cfg: -1 = tiles by shape; terms: 6 (default) or 9 (all cross terms: every f32 product exact)."""
    _chk(x)
    out = _empty((G, M, N) if G > 1 else (M, N), dtype=torch.float32, device=x.device) if out is None else out
    call("frcnn_gemm_x3", _ptr(x), _ptr(planes), _ptr(bias), _ptr(residual), _ptr(out), int(G), int(M), int(N), int(K), int(act),
         int(cfg), int(terms), _stream())
    return out


class H2(object):
    """A float32 tensor [rows, K] held as frcnn_gemm_h2 operand planes: `planes` uint8 buffer = fp16 [2][rows][K] (H then L) and
    `inv` f32 [K/128][rows] = the exact power-of-two block scales 2^-e (csrc/gemm_h2.hip)."""
    __slots__ = ("planes", "inv", "rows", "K")

    def __init__(self, planes, inv, rows, K):
        self.planes, self.inv, self.rows, self.K = planes, inv, int(rows), int(K)

    @staticmethod
    def empty(rows, K, device):
        assert K % 128 == 0
        return H2(torch.empty(lib().frcnn_h2_planes_bytes(int(rows), int(K)), dtype=torch.uint8, device=device),
                  torch.empty((K // 128, rows), dtype=torch.float32, device=device), rows, K)

    def to_float(self):
        """(h + l) * 2^-e as float32 [rows, K] (tests / debugging: host-side torch arithmetic, not a product path)."""
        hl = self.planes.view(torch.float16).view(2, self.rows, self.K).float()
        inv = self.inv.t().repeat_interleave(128, dim=1)
        return (hl[0] + hl[1]) * inv


def h2_split(x, out=None):
    """frcnn_h2_split: x f32 [..., K] (K % 128 == 0) -> H2 planes of its rows."""
    _chk(x)
    K = x.shape[-1]
    rows = x.numel() // K
    out = H2.empty(rows, K, x.device) if out is None else out
    assert out.rows == rows and out.K == K
    call("frcnn_h2_split", _ptr(x), rows, K, _ptr(out.planes), _ptr(out.inv), _stream())
    return out


def h2_pack_w(w, out=None):
    """frcnn_h2_pack_w: filter bank w [G,N,K] or [N,...K] f32 -> (planes uint8 [G][2][N][K] fp16, w_inv f32 [G,N])."""
    _chk(w)
    G, N, K = (w.shape[0], w.shape[1], w.shape[2]) if w.dim() == 3 else (1, w.shape[0], w[0].numel())
    if out is None:
        out = (torch.empty(lib().frcnn_h2_planes_bytes(G * N, K), dtype=torch.uint8, device=w.device),
               torch.empty((G, N), dtype=torch.float32, device=w.device))
    call("frcnn_h2_pack_w", _ptr(w), G, N, K, _ptr(out[0]), _ptr(out[1]), _stream())
    return out


def gemm_h2(x, wp, G, M, N, K, bias=None, residual=None, act=ACT_NONE, out=None, out_planes=None, want_f32=True, cfg=-1, mask=None):
    """out[g] = act(x[g] W[g]^T + bias + residual[g]) through frcnn_gemm_h2.  x: H2 of [G*M, K]; wp = h2_pack_w(W [G,N,K]).
    residual: f32 [G*M, N] or an H2 of [G*M, N] (read as (h + l) * 2^-e).  out: f32 [G*M, N] (allocated when want_f32 and out is None);
    out_planes: an H2 of [G*M, N] to ALSO receive the result as the next GEMM's operand (emitted from the epilogue).
    mask (training; float32 [G*M, N]): result = mask > 0 ? result : 0 in the tile epilogue (frcnn_gemm_h2_masked; float32 residual only).
    Returns (out or None, out_planes or None)."""
    assert isinstance(x, H2) and x.rows == G * M and x.K == K
    if out is None and want_f32:
        out = _empty((G * M, N), dtype=torch.float32, device=x.planes.device)
    if out_planes is not None:
        assert out_planes.rows == G * M and out_planes.K == N
    rp = residual if isinstance(residual, H2) else None
    if rp is not None:
        assert rp.rows == G * M and rp.K == N
    if mask is not None:
        _chk(mask)
        assert rp is None and mask.numel() == G * M * N
        call("frcnn_gemm_h2_masked", _ptr(x.planes), _ptr(x.inv), _ptr(wp[0]), _ptr(wp[1]), _ptr(bias), _ptr(residual), _ptr(mask), _ptr(out),
             _ptr(None if out_planes is None else out_planes.planes), _ptr(None if out_planes is None else out_planes.inv),
             int(G), int(M), int(N), int(K), int(act), int(cfg), _stream())
        return out, out_planes
    call("frcnn_gemm_h2", _ptr(x.planes), _ptr(x.inv), _ptr(wp[0]), _ptr(wp[1]), _ptr(bias), _ptr(None if rp is not None else residual),
         _ptr(None if rp is None else rp.planes), _ptr(None if rp is None else rp.inv), _ptr(out),
         _ptr(None if out_planes is None else out_planes.planes), _ptr(None if out_planes is None else out_planes.inv),
         int(G), int(M), int(N), int(K), int(act), int(cfg), _stream())
    return out, out_planes


def gemm_h2_mean(x, wp, G, M, N, K, bias, residual, act, rows, out=None, cfg=-1):
    """frcnn_gemm_h2_mean: [G * M / rows, N] = mean over each group of `rows` consecutive rows of act(x[g] W^T + bias + residual[g]); one
    batch entry per image (the reduction order is then independent of the batch slot).  residual: f32 [G*M, N], an H2, or None."""
    assert isinstance(x, H2) and x.rows == G * M and x.K == K and M % rows == 0
    if out is None:
        out = _empty((G * (M // rows), N), dtype=torch.float32, device=x.planes.device)
    rp = residual if isinstance(residual, H2) else None
    ws = workspace(lib().frcnn_gemm_h2_mean_workspace_bytes(int(G), int(M), int(N)), x.planes.device, "h2_mean")
    call("frcnn_gemm_h2_mean", _ptr(x.planes), _ptr(x.inv), _ptr(wp[0]), _ptr(wp[1]), _ptr(bias), _ptr(None if rp is not None else residual),
         _ptr(None if rp is None else rp.planes), _ptr(None if rp is None else rp.inv), int(G), int(M), int(N), int(K), int(act), int(rows),
         _ptr(out), _ptr(ws), ws.numel() * ws.element_size(), int(cfg), _stream())
    return out


def winograd_input_transform_h2(x, v, m):
    """x [N,H,W,C] f32 -> V as H2 planes of [points * tiles, C] (frcnn_winograd[7]_input_transform_h2)."""
    _chk(x)
    N, H, W, C = x.shape
    assert isinstance(v, H2) and v.K == C and v.rows == winograd_points(m) * winograd_tiles(N, H, W, m)
    if m == 7:
        call("frcnn_winograd7_input_transform_h2", _ptr(x), N, C, _ptr(v.planes), _ptr(v.inv), _stream())
    else:
        call("frcnn_winograd_input_transform_h2", _ptr(x), N, H, W, C, int(m), _ptr(v.planes), _ptr(v.inv), _stream())
    return v


def winograd_output_transform_h2(mm, bias, act, shape, m, out_planes, out=None):
    """mm [(points), tiles, C] f32 -> act(A^T M A + bias) of shape [N,H,W,C] as H2 planes (`out_planes`, rows = N*H*W) and, when `out`
    is given, as the float32 tensor too."""
    _chk(mm)
    N, H, W, C = shape
    assert isinstance(out_planes, H2) and out_planes.rows == N * H * W and out_planes.K == C
    assert mm.numel() == winograd_points(m) * winograd_tiles(N, H, W, m) * C
    if out is not None:
        _chk(out)
        assert tuple(out.shape) == tuple(shape)
    if m == 7:
        call("frcnn_winograd7_output_transform_h2", _ptr(mm), N, C, _ptr(bias), int(act), _ptr(out), _ptr(out_planes.planes),
             _ptr(out_planes.inv), _stream())
    else:
        call("frcnn_winograd_output_transform_h2", _ptr(mm), N, H, W, C, int(m), _ptr(bias), int(act), _ptr(out), _ptr(out_planes.planes),
             _ptr(out_planes.inv), _stream())
    return out_planes


def relu6_bwd(grad, y):
    _chk(grad), _chk(y)
    call("frcnn_relu6_bwd", _ptr(grad), _ptr(y), grad.numel(), _stream())
    return grad


def maxpool_bwd(x, y, dy, k, stride, dx):
    """gradient of maxpool(x) (padding bottom / right only) into dx (written)."""
    _chk(x), _chk(y), _chk(dy), _chk(dx)
    N, H, W, C = x.shape
    call("frcnn_maxpool_bwd", _ptr(x), N, H, W, C, int(k), int(stride), _ptr(y), _ptr(dy), y.shape[1], y.shape[2], _ptr(dx), _stream())
    return dx


def dropout(x, seed, keep_prob, out=None, step_mult=0):
    """tf.nn.dropout with a counter-based mask of (seed, element index); the same call on a gradient is the backward pass.
    step_mult: seed = step_mult * (the step's sampling seed) + const -- tells a recorded training step how to advance it."""
    _chk(x)
    out = _empty(tuple(x.shape), dtype=x.dtype, device=x.device) if out is None else out
    call("frcnn_dropout", _ptr(x), x.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, float(keep_prob), _ptr(out), _stream())
    if _binding.recorder is not None and step_mult:
        _binding.recorder.patch_last(2, step_mult)
    return out


def dwconv3x3_dgrad(g, w, stride, pad, dx, accumulate=False):
    _chk(g), _chk(w), _chk(dx)
    N, OH, OW, C = g.shape
    call("frcnn_dwconv3x3_dgrad", _ptr(g), N, OH, OW, C, _ptr(w), _ptr(dx), dx.shape[1], dx.shape[2], int(stride), int(pad[0]), int(pad[2]),
         1 if accumulate else 0, _stream())
    return dx


def dwconv3x3_wgrad(g, x, stride, pad, scale, dw):
    _chk(g), _chk(x), _chk(dw)
    N, OH, OW, C = g.shape
    ws = workspace(lib().frcnn_dwconv3x3_wgrad_workspace_bytes(N, OH, OW, C), g.device, "dw_wgrad")
    call("frcnn_dwconv3x3_wgrad", _ptr(g), _ptr(x), N, x.shape[1], x.shape[2], C, OH, OW, int(stride), int(pad[0]), int(pad[2]), _ptr(scale),
         _ptr(dw), _ptr(ws), ws.numel(), _stream())
    return dw


def dwconv3x3_refold(w, scale, wf):
    _chk(w), _chk(scale), _chk(wf)
    call("frcnn_dwconv3x3_refold", _ptr(w), _ptr(scale), w.shape[-1], _ptr(wf), _stream())
    return wf


def add_strided(src, dst, stride, accumulate):
    _chk(src), _chk(dst)
    N, OH, OW, C = src.shape
    call("frcnn_add_strided", _ptr(src), N, OH, OW, C, _ptr(dst), dst.shape[1], dst.shape[2], int(stride), 1 if accumulate else 0, _stream())
    return dst


def spatial_mean_bwd(dy, HW, out):
    _chk(dy), _chk(out)
    call("frcnn_spatial_mean_bwd", _ptr(dy), dy.shape[0], int(HW), dy.shape[1], _ptr(out), _stream())
    return out


def colsum(dy2d, out):
    _chk(dy2d), _chk(out)
    call("frcnn_colsum", _ptr(dy2d), dy2d.shape[0], dy2d.shape[1], _ptr(out), _stream())
    return out


def crop_and_resize_bwd(dout, rois, feat_stride, dfeat, max_lds=0):
    """dfeat += gradient of crop_and_resize w.r.t. the feature map.  max_lds (bytes; tests): the LDS budget of the launch plan."""
    _chk(dout), _chk(rois), _chk(dfeat)
    R, P, _, C = dout.shape
    H, W = dfeat.shape[-3], dfeat.shape[-2]
    if max_lds:
        call("frcnn_crop_and_resize_bwd_plan", _ptr(dout), H, W, C, _ptr(rois), R, float(feat_stride), P, _ptr(dfeat), int(max_lds), _stream())
    else:
        call("frcnn_crop_and_resize_bwd", _ptr(dout), H, W, C, _ptr(rois), R, float(feat_stride), P, _ptr(dfeat), _stream())
    return dfeat


def sgd_momentum(w, acc, w_folded, grad, scale, K, lr, momentum, weight_decay, grad_scale=1.0):
    call("frcnn_sgd_momentum", _ptr(w), _ptr(acc), _ptr(w_folded), _ptr(grad), _ptr(scale), w.numel(), int(K), float(lr),
         float(momentum), float(weight_decay), float(grad_scale), _stream())


def sgd_desc_table(entries, device):
    """entries: list of (w, acc, w_folded or None, grad, scale or None, K, lr_mult, weight_decay) tensors / numbers ->
    uint8 device tensor holding the SgdDesc array of frcnn_sgd_momentum_multi."""
    import struct
    nb = lib().frcnn_sgd_desc_bytes()
    assert nb == 64, nb
    raw = bytearray()
    for w, acc, wf, grad, scale, K, lr_mult, wd in entries:
        raw += struct.pack("<QQQQQqiffi", w.data_ptr(), acc.data_ptr(), 0 if wf is None else wf.data_ptr(), grad.data_ptr(),
                           0 if scale is None else scale.data_ptr(), int(w.numel()), int(K), float(lr_mult), float(wd), 0)
    return torch.frombuffer(raw, dtype=torch.uint8).clone().to(device)


def sgd_momentum_multi(table, count, lr, momentum, grad_scale=1.0):
    call("frcnn_sgd_momentum_multi", _ptr(table), int(count), float(lr), float(momentum), float(grad_scale), _stream())


def sgd_momentum_range(table, first, count, lr, momentum, grad_scale=1.0):
    """frcnn_sgd_momentum_multi for the descriptors [first, first + count) of `table`."""
    call("frcnn_sgd_momentum_range", _ptr(table), int(first), int(count), float(lr), float(momentum), float(grad_scale), _stream())


def sumsq(w, scale, out, accumulate):
    ws = workspace(4096, w.device, "sumsq")
    call("frcnn_sumsq", _ptr(w), w.numel(), float(scale), _ptr(out), 1 if accumulate else 0, _ptr(ws), ws.numel(), _stream())


def sumsq_multi(ptr_table, sizes, scale, out, accumulate=False):
    """ptr_table int64 [count] (device pointers of float tensors), sizes int64 [count], both on device."""
    count = ptr_table.numel()
    ws = workspace(8 * 64 * count, out.device, "sumsq")
    call("frcnn_sumsq_multi", _ptr(ptr_table), _ptr(sizes), count, float(scale), _ptr(out), 1 if accumulate else 0, _ptr(ws), ws.numel(),
         _stream())
    return out


class Graph:
    """One captured hipGraph (frcnn_graph_* in the C ABI)."""

    def __init__(self):
        self.handle = None

    def capture(self, fn):
        st = torch.cuda.current_stream()
        if st.cuda_stream == 0:
            raise RuntimeError("capture needs a non-default stream (use torch.cuda.stream(torch.cuda.Stream()))")
        call("frcnn_graph_begin", ctypes.c_void_p(st.cuda_stream))
        try:
            fn()
        finally:
            h = ctypes.c_void_p()
            call("frcnn_graph_end", ctypes.c_void_p(st.cuda_stream), ctypes.byref(h))
        self.handle = h
        return self

    def launch(self):
        call("frcnn_graph_launch", self.handle, _stream())

    def __del__(self):
        try:
            if self.handle:
                call("frcnn_graph_destroy", self.handle)
        except Exception:
            pass
