"""model.bbox_transform -- numpy-in / numpy-out box codec of the reference (lib/model/bbox_transform.py:14-81), executed
by libfrcnn_hip.so (frcnn_bbox_transform / frcnn_bbox_transform_inv / frcnn_clip_boxes).

Same signatures and conventions: boxes (x1, y1, x2, y2) with the +1 pixel width, deltas [N, 4k] strided 0::4 per class,
`clip_boxes` clamps every coordinate to [0, dim-1] (im_shape = (height, width, ...)) and, like the reference, works on
the array it is given and returns it.  float32 throughout (the reference casts boxes to the deltas' dtype, :40)."""
import numpy as np
import torch

from frcnn_hip import ops


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_dev())


def bbox_transform(ex_rois, gt_rois):
    ex_rois, gt_rois = np.asarray(ex_rois), np.asarray(gt_rois)
    if ex_rois.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.float32)
    return ops.bbox_transform(_t(ex_rois[:, :4]), _t(gt_rois[:, :4])).cpu().numpy()


def bbox_transform_inv(boxes, deltas):
    boxes, deltas = np.asarray(boxes), np.asarray(deltas)
    if boxes.shape[0] == 0:                                    # :36-37
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    return ops.bbox_transform_inv(_t(boxes[:, :4]), _t(deltas)).cpu().numpy().astype(deltas.dtype, copy=False)


def clip_boxes(boxes, im_shape):
    boxes = np.asarray(boxes)
    if boxes.shape[0] == 0:
        return boxes
    out = ops.clip_boxes(_t(boxes), float(im_shape[0]), float(im_shape[1])).cpu().numpy()
    if isinstance(boxes, np.ndarray) and boxes.dtype == np.float32 and boxes.flags.writeable:
        boxes[...] = out                                       # the reference clips in place (:72-80)
        return boxes
    return out.astype(boxes.dtype, copy=False)
