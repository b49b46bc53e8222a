"""frcnn_hip -- ctypes binding of libfrcnn_hip.so (include/frcnn_hip.h), the MI355X hot path.

PyTorch-ROCm tensors are used for device storage and streams only: every op below hands raw
device pointers + the current hipStream_t to the C ABI.  There is NO CPU / eager fallback: if the
library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libfrcnn_hip.so")
_lib = None

ABI_VERSION = 6                          # include/frcnn_hip.h FRCNN_ABI_VERSION: the signatures below are this version's
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
NMS_RULE_CPU, NMS_RULE_GPU = 0, 1        # FRCNN_NMS_RULE_*: `(double)ovr >= thresh` (cpu_nms.pyx:65) / `ovr > (float)thresh` (nms_kernel.cu:71)

# name -> (restype, argtypes); must list every symbol include/frcnn_hip.h declares
_P = c_void_p
SIGNATURES = {
    "frcnn_abi_version": (c_int, []),
    "frcnn_build_info": (ctypes.c_char_p, []),
    "_nms": (None, [_P, _P, _P, c_int, c_int, c_float, c_int]),
    "frcnn_nms_workspace_bytes": (c_size_t, [c_int]),
    "frcnn_nms": (c_int, [_P, c_int, c_double, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_nms_sorted": (c_int, [_P, c_int, c_int, c_double, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_nms_rule": (c_int, [_P, c_int, c_double, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_nms_sorted_rule": (c_int, [_P, c_int, c_int, c_double, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_generate_anchors": (c_int, [c_int, _P, c_int, _P, c_int, _P]),
    "frcnn_generate_anchors_pre": (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P]),
    "frcnn_proposal_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "frcnn_proposal_layer": (c_int, [_P, _P, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_double,
                                     _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_proposal_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "frcnn_proposal_layer_batched": (c_int, [_P, _P, c_int, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_double,
                                             c_int, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_proposal_top_layer_inds": (c_int, [_P, _P, c_float, c_float, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    "frcnn_proposal_layer_tf_batched": (c_int, [_P, _P, c_int, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P,
                                                _P, _P, c_size_t, _P]),
    "frcnn_proposal_top_layer": (c_int, [_P, _P, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P,
                                         c_size_t, _P]),
    "frcnn_non_max_suppression": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "frcnn_proposal_layer_tf": (c_int, [_P, _P, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, _P,
                                        _P, c_size_t, _P]),
    "frcnn_crop_and_resize": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_float, c_int, c_int, _P, _P]),
    "frcnn_crop_and_resize_batched": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_float, c_int, c_int, _P, _P]),
    "frcnn_crop_and_resize_bias_act": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_float, c_int, _P, c_int, _P, _P]),
    "frcnn_detect_set_tuning": (c_int, [c_int, c_int]),
    "frcnn_detect_post_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "frcnn_detect_post_batched": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_double, c_int, c_int, c_double, c_int, c_float, c_int,
                                          _P, _P, c_int, c_longlong, _P, c_size_t, _P]),
    "frcnn_detect_post_workspace_bytes": (c_size_t, [c_int, c_int]),
    "frcnn_detect_post": (c_int, [_P, _P, _P, _P, c_int, c_int, c_double, c_int, c_int, c_double, c_float, c_int, _P, _P,
                                  c_int, _P, c_size_t, _P]),
    "frcnn_im_detect_boxes": (c_int, [_P, _P, c_int, c_int, c_double, c_int, c_int, _P, _P]),
    "frcnn_bbox_overlaps": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "frcnn_bbox_transform_inv": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "frcnn_clip_boxes": (c_int, [_P, c_int, c_int, c_float, c_float, _P]),
    "frcnn_bbox_transform": (c_int, [_P, _P, c_int, _P, _P]),
    "frcnn_conv2d_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_conv2d_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "frcnn_conv2d_nhwc_ws": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "frcnn_conv2d_nhwc_masked_ws": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, _P, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "frcnn_gemm_h2_masked": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_winograd_output_transform_masked": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "frcnn_winograd7_output_transform_masked": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P]),
    "frcnn_crc32c": (ctypes.c_uint32, [_P, c_size_t, ctypes.c_uint32]),
    "frcnn_snappy_uncompress": (c_longlong, [_P, c_size_t, _P, c_size_t]),
    "frcnn_prep_image_shape": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P]),
    "frcnn_prep_image": (c_int, [_P, c_int, c_int, c_int, _P, c_double, _P, c_int, c_int, c_int, _P]),
    "frcnn_gemm_batched_nt": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "frcnn_winograd_filter_transform": (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    "frcnn_winograd_filter_transform_device": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_winograd_input_transform": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_winograd_output_transform": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "frcnn_winograd7_filter_transform": (c_int, [_P, c_int, c_int, _P, _P]),
    "frcnn_winograd7_filter_transform_device": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_winograd7_input_transform": (c_int, [_P, c_int, c_int, _P, _P]),
    "frcnn_winograd7_output_transform": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P]),
    "frcnn_winograd_input_transform_h2": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "frcnn_winograd_output_transform_h2": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "frcnn_winograd7_input_transform_h2": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "frcnn_winograd7_output_transform_h2": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "frcnn_set_tuning": (c_int, [c_int, c_int]),
    "frcnn_pack_filter_hwio": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_maxpool_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "frcnn_dwconv3x3_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_dwconv3x3_nhwc_h2": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_spatial_mean": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_softmax_rows": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_rpn_softmax": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_copy_cols": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "frcnn_anchor_target_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "frcnn_anchor_target_layer": (c_int, [_P, c_int, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_double, c_double,
                                          c_double, c_longlong, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_anchor_target_layer_inject": (c_int, [_P, c_int, c_float, c_float, c_int, c_int, c_int, c_int, _P, c_int, c_double, c_double,
                                                 c_double, _P, c_int, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_proposal_target_layer_inject": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frcnn_proposal_target_layer": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, _P, _P,
                                            c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frcnn_proposal_target_layer_dn": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, _P, _P,
                                               c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frcnn_loss_workspace_bytes": (c_size_t, [c_longlong]),
    "frcnn_softmax_ce_loss": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_smooth_l1_loss": (c_int, [_P, _P, _P, _P, c_longlong, c_float, c_float, _P, _P, _P, c_size_t, _P]),
    "frcnn_transpose_pad": (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    "frcnn_im2col_t": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "frcnn_flip_transpose_filter": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_conv2d_dgrad_strided": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int,
                                           c_int, c_int, _P]),
    "frcnn_conv2d_wgrad_supported": (c_int, [c_int, c_int]),
    "frcnn_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "frcnn_conv2d_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P,
                                   c_size_t, _P]),
    "frcnn_conv2d_wgrad_set_plan": (None, [c_int, c_int]),
    "frcnn_conv2d_wgrad_h2_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "frcnn_conv2d_wgrad_h2": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P,
                                      c_size_t, _P]),
    "frcnn_conv2d_wgrad_h2_set_plan": (None, [c_int, c_int]),
    "frcnn_relu_bwd": (c_int, [_P, _P, c_longlong, _P]),
    "frcnn_gemm_x3_pack_bytes": (c_size_t, [c_int, c_int, c_int]),
    "frcnn_gemm_x3_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_gemm_x3": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_h2_planes_bytes": (c_size_t, [c_longlong, c_int]),
    "frcnn_h2_pack_w": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "frcnn_h2_split": (c_int, [_P, c_longlong, c_int, _P, _P, _P]),
    "frcnn_gemm_h2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_gemm_h2_mean_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "frcnn_gemm_h2_mean": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, c_int, _P]),
    "frcnn_relu6_bwd": (c_int, [_P, _P, c_longlong, _P]),
    "frcnn_maxpool_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P]),
    "frcnn_dropout": (c_int, [_P, c_longlong, c_ulonglong, c_float, _P, _P]),
    "frcnn_dwconv3x3_dgrad": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "frcnn_dwconv3x3_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "frcnn_dwconv3x3_wgrad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_dwconv3x3_refold": (c_int, [_P, _P, c_int, _P, _P]),
    "frcnn_add_strided": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "frcnn_spatial_mean_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "frcnn_colsum": (c_int, [_P, c_int, c_int, _P, _P]),
    "frcnn_crop_and_resize_bwd": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_float, c_int, _P, _P]),
    "frcnn_crop_and_resize_bwd_plan": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_float, c_int, _P, c_size_t, _P]),
    "frcnn_sgd_momentum": (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, c_float, c_float, c_float, c_float, _P]),
    "frcnn_sgd_desc_bytes": (c_size_t, []),
    "frcnn_sgd_momentum_multi": (c_int, [_P, c_int, c_float, c_float, c_float, _P]),
    "frcnn_sgd_momentum_range": (c_int, [_P, c_int, c_int, c_float, c_float, c_float, _P]),
    "frcnn_sumsq": (c_int, [_P, c_longlong, c_double, _P, c_int, _P, c_size_t, _P]),
    "frcnn_sumsq_multi": (c_int, [_P, _P, c_int, c_double, _P, c_int, _P, c_size_t, _P]),
    "frcnn_graph_begin": (c_int, [_P]),
    "frcnn_graph_end": (c_int, [_P, ctypes.POINTER(c_void_p)]),
    "frcnn_graph_launch": (c_int, [_P, _P]),
    "frcnn_graph_destroy": (c_int, [_P]),
}


class FrcnnHipError(RuntimeError):
    pass


def lib():
    """Load libfrcnn_hip.so (once).  Raises ImportError loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libfrcnn_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or python tf-faster-rcnn_amd/frcnn_hip/build.py); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing: loud by design
            fn.restype, fn.argtypes = res, args
        got = L.frcnn_abi_version()
        if got != ABI_VERSION:             # a stale .so would take shifted arguments silently
            raise ImportError("libfrcnn_hip.so has ABI version %d, this binding is written for %d: rebuild (python -m frcnn_hip.build)"
                              % (got, ABI_VERSION))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        if rc <= -1000:
            raise FrcnnHipError("%s: HIP error %d" % (what, -rc - 1000))
        raise FrcnnHipError("%s: %s" % (what, {-1: "bad argument", -2: "workspace too small",
                                               -3: "shape not supported by the kernels"}.get(rc, "error %d" % rc)))


recorder = None                          # a replay.Recording while ONE eager training step is being recorded (frcnn_hip/replay.py), else None


def call(name, *args):
    fn = getattr(lib(), name)
    check(fn(*args), name)
    if recorder is not None:
        recorder.add_call(fn, name, args)
