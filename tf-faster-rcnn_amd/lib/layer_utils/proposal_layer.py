"""layer_utils.proposal_layer -- the tf.py_func seam of lib/nets/network.py:123-126, same signature
as lib/layer_utils/proposal_layer.py:16-53, executed by frcnn_proposal_layer.

Accepts numpy arrays (returns numpy, exactly like the reference) or CUDA torch tensors (returns the
zero-padded device buffers + the count, no host hop)."""
import numpy as np
import torch

from frcnn_hip import NMS_RULE_CPU, NMS_RULE_GPU, ops
from model.config import cfg


def proposal_layer(rpn_cls_prob, rpn_bbox_pred, im_info, cfg_key, _feat_stride, anchors, num_anchors):
    if type(cfg_key) == bytes:
        cfg_key = cfg_key.decode('utf-8')
    pre_nms_topN = cfg[cfg_key].RPN_PRE_NMS_TOP_N
    post_nms_topN = cfg[cfg_key].RPN_POST_NMS_TOP_N
    nms_thresh = cfg[cfg_key].RPN_NMS_THRESH
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    on_device = torch.is_tensor(rpn_cls_prob)
    dev = rpn_cls_prob.device if on_device else torch.device("cuda", torch.cuda.current_device())
    to_dev = (lambda a, dt=torch.float32: a.to(dev, dt).contiguous()) if on_device else \
        (lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt))
    # anchors[0:A] are the base anchors (shift 0); the kernel regenerates the rest from the index
    base = anchors[:num_anchors]
    base_d = (base if torch.is_tensor(base) else torch.from_numpy(np.ascontiguousarray(base))).to(dev, torch.float64).contiguous()
    info = im_info.cpu().numpy() if torch.is_tensor(im_info) else np.asarray(im_info)
    rois, scores, num = ops.proposal_layer(to_dev(rpn_cls_prob), to_dev(rpn_bbox_pred), float(info[0]), float(info[1]), stride,
                                           base_d, int(pre_nms_topN), int(post_nms_topN), float(nms_thresh),
                                           rule=NMS_RULE_GPU if cfg.USE_GPU_NMS else NMS_RULE_CPU)      # nms_wrapper.py:15-23
    if on_device:
        return rois, scores, num
    n = int(num.item())
    return rois[:n].cpu().numpy(), scores[:n].cpu().numpy()
