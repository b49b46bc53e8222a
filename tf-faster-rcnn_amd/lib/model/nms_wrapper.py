"""model.nms_wrapper -- the `nms(dets, thresh, force_cpu=False)` entry of the reference
(/root/reference/lib/model/nms_wrapper.py:15-23), re-hosted on libfrcnn_hip.so.

Contract kept: `dets` float32 [N,5] = (x1, y1, x2, y2, score); returns the list of kept ORIGINAL row indices in
descending-score order; an empty input returns [] without touching the device.  `cfg.USE_GPU_NMS` and `force_cpu`
choose between the two module names exactly like the reference does; both run the same HIP kernels and differ in the
suppression rule the reference's two implementations have: gpu_nms `ovr > thresh` in float32 (nms_kernel.cu:71),
cpu_nms `(double)ovr >= thresh` (cpu_nms.pyx:65).  There is no host implementation in the product."""
from model.config import cfg
import nms.cpu_nms as _cpu_mod
import nms.gpu_nms as _gpu_mod


def _pick_backend(force_cpu):
    use_accel_name = bool(cfg.USE_GPU_NMS) and not force_cpu
    if use_accel_name:
        return lambda d, t: _gpu_mod.gpu_nms(d, t, device_id=0)
    return _cpu_mod.cpu_nms


def nms(dets, thresh, force_cpu=False):
    n_boxes = int(dets.shape[0])
    if n_boxes == 0:            # nothing to suppress: same early-out as the reference
        return []
    run = _pick_backend(force_cpu)
    return run(dets, thresh)
