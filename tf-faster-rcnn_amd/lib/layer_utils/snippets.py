"""layer_utils.snippets.generate_anchors_pre -- lib/layer_utils/snippets.py:14-30 on the GPU."""
import numpy as np
import torch

from frcnn_hip import ops


def generate_anchors_pre(height, width, feat_stride, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
    """-> (anchors float32 [H*W*A, 4] numpy, length int32); anchor index (y*W + x)*A + a."""
    base = ops.generate_anchors(16, np.asarray(anchor_ratios, dtype=np.float64), np.asarray(anchor_scales, dtype=np.float64))
    dev = torch.device("cuda", torch.cuda.current_device())
    stride = int(np.asarray(feat_stride).reshape(-1)[0])
    anchors = ops.generate_anchors_pre(int(height), int(width), stride, torch.from_numpy(base).to(dev)).cpu().numpy()
    return anchors, np.int32(anchors.shape[0])
