"""Measurement builds: libfrcnn_hip.so with -DFRCNN_ABLATION (the extra frcnn_gemm_h2 / frcnn_gemm_x3 configurations the sweeps
compare; some give wrong results by construction) into /tmp, and frcnn_hip bound to it.  Import BEFORE frcnn_hip:

    import ablation_lib; ablation_lib.use()
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-faster-rcnn_amd")


def use(extra=()):
    sys.path[:0] = [PKG]
    from frcnn_hip import build as B
    so = "/tmp/libfrcnn_hip_ablation.so"
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DFRCNN_ABLATION"] + list(extra) + ["-shared", "-o", so] + srcs)
    import frcnn_hip
    frcnn_hip.LIB_PATH = so
    return so
