"""layer_utils.proposal_top_layer -- lib/layer_utils/proposal_top_layer.py:17-55 (TEST.MODE 'top') by
frcnn_proposal_top_layer.  Maps with fewer than RPN_TOP_N anchors take the reference's random fill (:30-33):
`npr.choice(length, size=rpn_top_n, replace=True)` is drawn HERE from numpy's global stream -- the same call the
reference makes, so a seeded run picks the same anchors -- and frcnn_proposal_top_layer_inds decodes those."""
import numpy as np
import numpy.random as npr
import torch

from frcnn_hip import ops
from model.config import cfg


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, im_info, _feat_stride, anchors, num_anchors):
    rpn_top_n = int(cfg.TEST.RPN_TOP_N)
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    base = t(np.asarray(anchors)[:num_anchors], torch.float64)
    length = int(rpn_cls_prob.shape[1] * rpn_cls_prob.shape[2] * num_anchors)
    if length < rpn_top_n:
        # Random selection, maybe unnecessary and loses good proposals, but such a case rarely happens (:31-33)
        top_inds = npr.choice(length, size=rpn_top_n, replace=True)
        rois, scores = ops.proposal_top_layer_inds(t(rpn_cls_prob), t(rpn_bbox_pred), float(im_info[0]), float(im_info[1]), stride,
                                                   base, t(top_inds.astype(np.int32), torch.int32))
    else:
        rois, scores = ops.proposal_top_layer(t(rpn_cls_prob), t(rpn_bbox_pred), float(im_info[0]), float(im_info[1]), stride,
                                              base, rpn_top_n)
    return rois.cpu().numpy(), scores.cpu().numpy()
