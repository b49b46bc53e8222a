"""utils.cython_bbox.bbox_overlaps -- drop-in for lib/utils/bbox.pyx:15-55 (float64 IoU matrix with the
+1 pixel convention), computed by frcnn_bbox_overlaps on the GPU."""
import numpy as np
import torch

from frcnn_hip import ops


def bbox_overlaps(boxes, query_boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64)
    if boxes.ndim != 2 or query_boxes.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")   # what the Cython buffer check raises
    if boxes.shape[0] == 0 or query_boxes.shape[0] == 0:
        return np.zeros((boxes.shape[0], query_boxes.shape[0]), dtype=np.float64)
    dev = torch.device("cuda", torch.cuda.current_device())
    out = ops.bbox_overlaps(torch.from_numpy(boxes[:, :4].copy()).to(dev), torch.from_numpy(query_boxes[:, :4].copy()).to(dev))
    return out.cpu().numpy()
