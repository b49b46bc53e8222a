"""model.nms_wrapper.nms -- same dispatcher as lib/model/nms_wrapper.py:15-23."""
from model.config import cfg
from nms.gpu_nms import gpu_nms
from nms.cpu_nms import cpu_nms


def nms(dets, thresh, force_cpu=False):
    """Dispatch to either CPU or GPU NMS implementations (both names run on the MI355X here)."""
    if dets.shape[0] == 0:
        return []
    if cfg.USE_GPU_NMS and not force_cpu:
        return gpu_nms(dets, thresh, device_id=0)
    return cpu_nms(dets, thresh)
