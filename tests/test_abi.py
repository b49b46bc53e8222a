"""CPU: the C-ABI library loads and exports every symbol include/frcnn_hip.h declares; host-side
entry points (no GPU needed) behave; the binding table matches the header."""
import ctypes
import os
import re

import numpy as np

import frcnn_oracle as ora

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+|_nms)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import frcnn_hip
    L = ctypes.CDLL(frcnn_hip.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
    assert sorted(frcnn_hip.SIGNATURES) == syms            # binding table == header
    declared = int(re.search(r"#define\s+FRCNN_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()).group(1))
    assert frcnn_hip.lib().frcnn_abi_version() == declared == frcnn_hip.ABI_VERSION == 6
    assert b"gfx950" in frcnn_hip.lib().frcnn_build_info()


def test_library_exports_the_reference_mangled_nms_name():
    """lib/nms/gpu_nms.hpp:1-2 declares `_nms` with C++ linkage and lib/nms/gpu_nms.pyx:13-14 binds it that way: a prebuilt gpu_nms
    extension of the reference imports `_Z4_nmsPiS_PKfiifi` (the very name tests/test_detect_gpu.py binds in the oracle build of the
    reference's nms_kernel.cu).  The library exports it beside the extern "C" name, so that extension relinks unchanged."""
    import subprocess
    import frcnn_hip
    L = ctypes.CDLL(frcnn_hip.LIB_PATH)
    assert hasattr(L, "_Z4_nmsPiS_PKfiifi") and hasattr(L, "_nms")
    out = subprocess.run(["c++filt", "_Z4_nmsPiS_PKfiifi"], capture_output=True, text=True).stdout.strip()
    if out:                                                # (binutils present) the name demangles to the reference's declaration
        assert out.replace(" ", "") == "_nms(int*,int*,floatconst*,int,int,float,int)"
    # no device: both names return num_out = 0 through their only error channel
    keep, n = np.zeros(4, dtype=np.int32), ctypes.c_int(7)
    boxes = np.zeros((4, 5), dtype=np.float32)
    fn = L._Z4_nmsPiS_PKfiifi
    fn.restype = None
    import torch
    if not torch.cuda.is_available():
        fn(keep.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n), boxes.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(4), ctypes.c_int(5),
           ctypes.c_float(0.5), ctypes.c_int(0))
        assert n.value == 0


def test_host_generate_anchors_matches_oracle():
    from frcnn_hip import ops
    for scales in ((8, 16, 32), (4, 8, 16, 32), (2, 4, 8, 16, 32)):
        for ratios in ((0.5, 1, 2), (0.3, 1.0, 3.1)):
            assert np.array_equal(ops.generate_anchors(16, ratios, scales), ora.generate_anchors(16, ratios, scales))


def test_host_pack_filter():
    from frcnn_hip import ops
    rng = np.random.RandomState(0)
    w = rng.randn(3, 3, 8, 5).astype(np.float32)
    s = rng.rand(5).astype(np.float32)
    p = ops.pack_filter_hwio(w, s)
    assert p.shape == (5, 3, 3, 8)
    assert np.array_equal(p, np.transpose(w * s[None, None, None, :], (3, 0, 1, 2)))
    assert np.array_equal(ops.pack_filter_hwio(w), np.transpose(w, (3, 0, 1, 2)))


def test_error_codes_without_gpu():
    import frcnn_hip
    L = frcnn_hip.lib()
    assert L.frcnn_generate_anchors(16, None, 3, None, 3, None) == -1
    assert L.frcnn_nms_workspace_bytes(6000) > 6000 * 94 * 8
    assert L.frcnn_proposal_workspace_bytes(38, 63, 9, 6000) > 0
    assert L.frcnn_detect_post_workspace_bytes(300, 21) > 0
    assert L.frcnn_conv2d_nhwc(None, 1, 1, 1, 32, None, None, None, 0, 0, 1, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, None) == -1
