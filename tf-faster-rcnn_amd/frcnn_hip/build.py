"""Builds libfrcnn_hip.so (gfx950) from tf-faster-rcnn_amd/csrc/*.hip with hipcc, in-tree.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc cross-compiles
without a GPU.  `-ffp-contract=off`: the detection kernels must keep numpy's one-rounding-per-op
arithmetic ("No FMA contraction"); the conv kernels do their math on MFMA and
are unaffected."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(PKG))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libfrcnn_hip.so")
SOURCES = ["detect_kernels.hip", "conv_igemm.hip", "dense_misc.hip", "train_kernels.hip", "backward_kernels.hip", "winograd.hip", "winograd7.hip", "preproc.hip", "gemm_x3.hip", "gemm_h2.hip", "wgrad_tn.hip", "wgrad_h2.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    # every header a translation unit may include: the C ABI + all of csrc/*.h (h2_common.h holds the operand-format rule that the
    # GEMM, the weight-gradient kernel, the splitter and the Winograd transforms must derive identically -- never link a mix)
    hdrs = [os.path.join(ROOT, "include", "frcnn_hip.h")] + sorted(glob.glob(os.path.join(CSRC, "*.h")))
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
