"""layer_utils.proposal_top_layer -- lib/layer_utils/proposal_top_layer.py:17-55 (TEST.MODE 'top') by
frcnn_proposal_top_layer.  The reference's random fill for maps with fewer than RPN_TOP_N anchors
(:30-33) is RNG dependent and not provided: FRCNN_E_UNSUPPORTED is raised instead."""
import numpy as np
import torch

from frcnn_hip import ops
from model.config import cfg


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, im_info, _feat_stride, anchors, num_anchors):
    rpn_top_n = cfg.TEST.RPN_TOP_N
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    rois, scores = ops.proposal_top_layer(t(rpn_cls_prob), t(rpn_bbox_pred), float(im_info[0]), float(im_info[1]), stride,
                                          t(anchors[:num_anchors], torch.float64), int(rpn_top_n))
    return rois.cpu().numpy(), scores.cpu().numpy()
