"""nms.cpu_nms -- import-compatible stand-in for lib/nms/cpu_nms.pyx:17-68.

lib/model/nms_wrapper.py:12-13 imports both native modules unconditionally, so the name must exist.
There is no CPU implementation in the product (the CPU restatement lives in oracle/ and is test
infrastructure only): `cpu_nms` executes the same HIP kernel as `gpu_nms`, whose suppression rule
already IS the cpu_nms rule."""
from nms.gpu_nms import gpu_nms


def cpu_nms(dets, thresh):
    return gpu_nms(dets, thresh, device_id=0)
