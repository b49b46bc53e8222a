"""Measurement builds: libfrcnn_hip.so with -DFRCNN_ABLATION (the extra frcnn_gemm_h2 / frcnn_gemm_x3 configurations the sweeps
compare; some give wrong results by construction) into /tmp, and frcnn_hip bound to it.  Import BEFORE frcnn_hip:

    import ablation_lib; ablation_lib.use()
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-faster-rcnn_amd")


PREBUILT = os.path.join(ROOT, "scratch", "libfrcnn_hip_ablation.so")      # built here by prebuild() (git-ignored, travels with the gpurun snapshot)


def prebuild(extra=(), out=None, ablation=True):
    """The ablation library compiled in-tree, one hipcc per translation unit in parallel (on the CPU box: the GPU box then loads it
    instead of spending minutes of GPU time in the compiler)."""
    sys.path[:0] = [PKG]
    from concurrent.futures import ThreadPoolExecutor
    from frcnn_hip import build as B
    out = PREBUILT if out is None else out
    obj_dir = os.path.join(B.CSRC, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in B.SOURCES]
    cmds = [[B._hipcc()] + B.FLAGS + (["-DFRCNN_ABLATION"] if ablation else []) + list(extra) + ["-c", os.path.join(B.CSRC, s), "-o", o] for s, o in zip(B.SOURCES, objs)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(subprocess.check_call, cmds))
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def use(extra=()):
    sys.path[:0] = [PKG]
    from frcnn_hip import build as B
    if not extra and os.path.exists(PREBUILT) and os.path.getmtime(PREBUILT) >= max(os.path.getmtime(os.path.join(B.CSRC, s)) for s in B.SOURCES):
        import frcnn_hip
        frcnn_hip.LIB_PATH = PREBUILT
        return PREBUILT
    so = "/tmp/libfrcnn_hip_ablation.so"
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DFRCNN_ABLATION"] + list(extra) + ["-shared", "-o", so] + srcs)
    import frcnn_hip
    frcnn_hip.LIB_PATH = so
    return so


if __name__ == "__main__":
    # python scratch/ablation_lib.py                      -> scratch/libfrcnn_hip_ablation.so (-DFRCNN_ABLATION)
    # python scratch/ablation_lib.py NAME -DFLAG [...]    -> scratch/libfrcnn_hip_NAME.so with the flags, no ablation configurations
    if len(sys.argv) > 1:
        print(prebuild(extra=sys.argv[2:], out=os.path.join(ROOT, "scratch", "libfrcnn_hip_%s.so" % sys.argv[1]), ablation=False))
    else:
        print(prebuild())
