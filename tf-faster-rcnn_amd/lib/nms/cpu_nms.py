"""nms.cpu_nms -- import-compatible stand-in for lib/nms/cpu_nms.pyx:17-68.

lib/model/nms_wrapper.py:12-13 imports both native modules unconditionally, so the name must exist.
There is no CPU implementation in the product (the CPU restatement lives in oracle/ and is test
infrastructure only): `cpu_nms` executes the HIP kernels with the Cython rule -- suppress iff
`(double)ovr >= thresh` (cpu_nms.pyx:65, cpu_nms.c:2239-2241) -- the path BASELINE.json pins."""
from frcnn_hip import NMS_RULE_CPU
from nms.gpu_nms import _run


def cpu_nms(dets, thresh):
    return _run(dets, float(thresh), 0, NMS_RULE_CPU)
