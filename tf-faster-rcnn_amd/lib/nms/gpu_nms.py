"""nms.gpu_nms -- drop-in for the reference's Cython module (lib/nms/gpu_nms.pyx:16-31).

Same signature and return value (`gpu_nms(dets f32[N,5], thresh, device_id=0)` -> list of kept
ORIGINAL indices in descending-score order).  Runs the HIP bitmask NMS of libfrcnn_hip.so with the
device-side greedy reduce and the rule of the reference's CUDA kernel: suppress iff
`devIoU > nms_overlap_thresh` in float32 (lib/nms/nms_kernel.cu:71; the threshold is a C float,
gpu_nms.pyx:16,30) -- so it returns what the reference `gpu_nms` returns, also at IoU == thresh."""
import numpy as np
import torch

from frcnn_hip import NMS_RULE_GPU, ops


def _run(dets, thresh, device_id, rule):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % dets.ndim)   # the Cython buffer check
    if dets.shape[0] == 0:
        return []
    dev = torch.device("cuda", int(device_id))
    with torch.cuda.device(dev):
        keep, num = ops.nms(torch.from_numpy(dets[:, :5].copy()).to(dev), float(thresh), rule=rule)
        n = int(num.item())
        return keep[:n].cpu().numpy().astype(np.intp).tolist()


def gpu_nms(dets, thresh, device_id=0):
    return _run(dets, np.float32(thresh), device_id, NMS_RULE_GPU)
