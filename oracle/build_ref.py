#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the reference's own native code into oracle/_ref/.

Recipe: the two first-party Cython sources of the reference
(/root/reference/lib/nms/cpu_nms.pyx, /root/reference/lib/utils/bbox.pyx) are compiled
*from where they lie* with a type-alias-only patch applied to a scratch copy under /tmp
(np.int_t -> np.intp_t, `np.float thresh` -> `double thresh`, np.float -> np.float64;
Cython 3 / numpy 2 no longer know the removed aliases).  The CUDA suppression kernel lib/nms/nms_kernel.cu is translated by
hipify-perl and cross-compiled for gfx950 (build_gpu_nms below).  Only the resulting .so files are
written into oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
No reference source is copied into this repository.

Usage: python oracle/build_ref.py            (no-op when /root/reference is absent)
"""
import os, re, shutil, subprocess, sys, tempfile, glob

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FRCNN_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")

SETUP = r'''
import numpy as np
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [Extension("nms.cpu_nms", ["nms/cpu_nms.pyx"], include_dirs=[np.get_include()],
                  extra_compile_args=["-O2", "-Wno-cpp", "-Wno-unused-function"]),
        Extension("utils.cython_bbox", ["utils/bbox.pyx"], include_dirs=[np.get_include()],
                  extra_compile_args=["-O2", "-Wno-cpp", "-Wno-unused-function"])]
setup(ext_modules=cythonize(exts, language_level=2, quiet=True))
'''


def build(force=False):
    if not os.path.isdir(os.path.join(REF, "lib", "nms")):
        return False
    have = glob.glob(os.path.join(OUT, "nms", "cpu_nms*.so")) and \
        glob.glob(os.path.join(OUT, "utils", "cython_bbox*.so"))
    if have and not force:
        return True
    work = tempfile.mkdtemp(prefix="frcnn_ref_")
    try:
        os.makedirs(os.path.join(work, "nms")); os.makedirs(os.path.join(work, "utils"))
        src = open(os.path.join(REF, "lib/nms/cpu_nms.pyx")).read()
        src = src.replace("np.int_t", "np.intp_t").replace("np.float thresh", "double thresh")
        src = src.replace("dtype=np.int)", "dtype=np.intp)")
        open(os.path.join(work, "nms/cpu_nms.pyx"), "w").write(src)
        src = open(os.path.join(REF, "lib/utils/bbox.pyx")).read()
        src = re.sub(r"^DTYPE = np\.float$", "DTYPE = np.float64", src, flags=re.M)
        src = src.replace("ctypedef np.float_t DTYPE_t", "ctypedef np.float64_t DTYPE_t")
        open(os.path.join(work, "utils/bbox.pyx"), "w").write(src)
        open(os.path.join(work, "setup.py"), "w").write(SETUP)
        subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=work,
                              stdout=subprocess.DEVNULL)
        for sub in ("nms", "utils"):
            os.makedirs(os.path.join(OUT, sub), exist_ok=True)
            for so in glob.glob(os.path.join(work, sub, "*.so")):
                shutil.copy(so, os.path.join(OUT, sub))
        # package markers (ours, not the reference's)
        open(os.path.join(OUT, "nms", "__init__.py"), "w").write(
            "import os\n_r = os.environ.get('FRCNN_REFERENCE', '/root/reference')\n"
            "if os.path.isdir(_r + '/lib/nms'):\n    __path__.append(_r + '/lib/nms')\n")
        open(os.path.join(OUT, "nms", "gpu_nms.py"), "w").write(
            "def gpu_nms(*a, **k):\n    raise RuntimeError('no CUDA in the oracle')\n")
        open(os.path.join(OUT, "utils", "__init__.py"), "w").write(
            "import os\n_r = os.environ.get('FRCNN_REFERENCE', '/root/reference')\n"
            "if os.path.isdir(_r + '/lib/utils'):\n    __path__.append(_r + '/lib/utils')\n")
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return True


def build_gpu_nms(force=False):
    """The reference's CUDA suppression kernel (lib/nms/nms_kernel.cu:24-144: devIoU, nms_kernel, the host loop _nms) as a gfx950
    library: `hipify-perl` translates the file where it lies (runtime API names only; the kernel body is untouched), hipcc
    cross-compiles it.  Two builds, because nvcc contracts a*b+c into FMAs by default and the rounding of Sa + Sb - interS depends on it:
    libref_gpu_nms.so (-ffp-contract=off, separate roundings like lib/nms/py_cpu_nms.py) and libref_gpu_nms_fma.so (-ffp-contract=fast).
    Entry: _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh, int device_id),
    C++-mangled `_Z4_nmsPiS_PKfiifi`.  tests/test_detect_gpu.py pins FRCNN_NMS_RULE_GPU against it on the GPU box."""
    src = os.path.join(REF, "lib", "nms", "nms_kernel.cu")
    hipify, hipcc = shutil.which("hipify-perl") or "/opt/rocm/bin/hipify-perl", shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not (os.path.isfile(src) and os.path.exists(hipify) and os.path.exists(hipcc)):
        return False
    outs = [os.path.join(OUT, "nms", "libref_gpu_nms.so"), os.path.join(OUT, "nms", "libref_gpu_nms_fma.so")]
    if all(os.path.exists(o) for o in outs) and not force:
        return True
    work = tempfile.mkdtemp(prefix="frcnn_ref_gpu_")
    try:
        hip = subprocess.run([hipify, src], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        open(os.path.join(work, "nms_kernel.hip"), "wb").write(hip)
        os.makedirs(os.path.join(OUT, "nms"), exist_ok=True)
        for out, contract in zip(outs, ("off", "fast")):
            subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=" + contract, "-include", "cstring",
                                   "-I" + os.path.join(REF, "lib", "nms"), os.path.join(work, "nms_kernel.hip"), "-o", out],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    if ok:
        print("reference CUDA NMS kernel (hipified) built" if build_gpu_nms(force="--force" in sys.argv) else "reference CUDA NMS kernel: not built")
    print("oracle/_ref built" if ok else "reference tree not present: nothing built")
