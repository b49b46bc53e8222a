"""nms.gpu_nms -- drop-in for the reference's Cython module (lib/nms/gpu_nms.pyx:16-31).

Same signature and return value (`gpu_nms(dets f32[N,5], thresh, device_id=0)` -> list of kept
ORIGINAL indices in descending-score order).  Runs the HIP bitmask NMS of libfrcnn_hip.so with the
device-side greedy reduce; suppression rule is the CPU/Cython one (`ovr >= thresh`,
lib/nms/cpu_nms.pyx:65) -- the path this project pins parity on."""
import numpy as np
import torch

from frcnn_hip import ops


def gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    dev = torch.device("cuda", int(device_id))
    with torch.cuda.device(dev):
        keep, num = ops.nms(torch.from_numpy(dets[:, :5].copy()).to(dev), float(thresh))
        n = int(num.item())
        return keep[:n].cpu().numpy().astype(np.intp).tolist()
