"""layer_utils.generate_anchors -- lib/layer_utils/generate_anchors.py:41-52 through the C ABI
(frcnn_generate_anchors, host function: float64, np.round half-to-even)."""
import numpy as np

from frcnn_hip import ops


def generate_anchors(base_size=16, ratios=[0.5, 1, 2], scales=2 ** np.arange(3, 6)):
    """Anchor (reference) windows: aspect ratios X scales wrt a (0, 0, 15, 15) window; [A,4] float64,
    ratio-major."""
    return ops.generate_anchors(base_size, np.asarray(ratios, dtype=np.float64), np.asarray(scales, dtype=np.float64))
