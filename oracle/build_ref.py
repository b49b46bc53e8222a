#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the reference's own native code into oracle/_ref/.

Recipe: the two first-party Cython sources of the reference
(/root/reference/lib/nms/cpu_nms.pyx, /root/reference/lib/utils/bbox.pyx) are compiled
*from where they lie* with a type-alias-only patch applied to a scratch copy under /tmp
(np.int_t -> np.intp_t, `np.float thresh` -> `double thresh`, np.float -> np.float64;
Cython 3 / numpy 2 no longer know the removed aliases).  Only the resulting .so files are
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


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built" if ok else "reference tree not present: nothing built")
