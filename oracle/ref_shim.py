"""TEST INFRASTRUCTURE ONLY.  Import the *reference's own* first-party Python (numpy/Cython path)
from /root/reference without TensorFlow / easydict / CUDA (recipe: SURVEY.md section A.7).

Only usable in the build container (where /root/reference exists).  Nothing on the product path,
in smoke() or in the `-m gpu` tests imports this module; it exists to (a) generate the golden
fixtures under tests/golden/ (oracle/gen_golden.py) and (b) pin oracle/frcnn_oracle.py against the
reference in the CPU test-suite.

The reference's package names (`model`, `nms`, `utils`, `layer_utils`) collide with this repo's
drop-in host mirror, therefore load_reference() must run in a process that has NOT imported the
mirror (gen_golden.py and the pin test use a subprocess).
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("FRCNN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available():
    return os.path.isdir(os.path.join(REF, "lib", "layer_utils"))


class _EasyDict(dict):
    """10-line stand-in for easydict.EasyDict (lib/model/config.py:9 imports it)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def load_reference():
    """Returns a namespace holding the reference's functions (numpy/Cython path, USE_GPU_NMS off)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF)
    from build_ref import build  # oracle/build_ref.py
    build()
    # removed numpy aliases still used by the reference (anchor_target_layer.py:48-49 ...)
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    if "tensorflow" not in sys.modules:
        sys.modules["tensorflow"] = types.ModuleType("tensorflow")  # imported, never called
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _EasyDict
        sys.modules["easydict"] = m
    for p in (os.path.join(REF, "lib"), os.path.join(HERE, "_ref")):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [os.path.join(HERE, "_ref"), os.path.join(REF, "lib")]

    from model.config import cfg
    cfg.USE_GPU_NMS = False
    from layer_utils.generate_anchors import generate_anchors
    from layer_utils.snippets import generate_anchors_pre, generate_anchors_pre_tf
    from layer_utils.proposal_layer import proposal_layer, proposal_layer_tf
    from layer_utils.proposal_top_layer import proposal_top_layer, proposal_top_layer_tf
    from layer_utils.anchor_target_layer import anchor_target_layer
    from layer_utils.proposal_target_layer import proposal_target_layer
    from model.bbox_transform import bbox_transform, bbox_transform_inv, clip_boxes
    from model.nms_wrapper import nms
    from nms.cpu_nms import cpu_nms
    from nms.py_cpu_nms import py_cpu_nms
    from utils.cython_bbox import bbox_overlaps

    ns = types.SimpleNamespace(**{k: v for k, v in locals().items() if callable(v) or k == "cfg"})
    return ns


def install_tf_numpy(ora):
    """Give the stub `tensorflow` module the numpy-backed ops of oracle/tf_numpy_shim.py so the reference's *_tf function
    bodies (USE_E2E_TF graph) can run; `ora` supplies the two restated TensorFlow kernels."""
    import tf_numpy_shim
    sys.modules["tensorflow"].__dict__.update({k: v for k, v in tf_numpy_shim.build(ora).__dict__.items() if not k.startswith("__")})
