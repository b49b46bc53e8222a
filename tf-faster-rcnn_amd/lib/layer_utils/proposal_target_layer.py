"""layer_utils.proposal_target_layer -- the tf.py_func seam of lib/nets/network.py:187-191 with the signature and outputs
of lib/layer_utils/proposal_target_layer.py:18-152, computed by libfrcnn_hip.so.

"Host-oracle" sampling like layer_utils.anchor_target_layer: the fg / bg RoIs are drawn with the reference's own
`npr.choice` calls on numpy's global stream (:119-135, same candidate arrays, sizes, `replace` flags and order); the IoU
matrix (frcnn_bbox_overlaps, float64), the gt assignment, labels, normalised class-expanded regression targets and weights
(frcnn_proposal_target_layer_inject) are device work.  TRAIN.USE_GT, TRAIN.BBOX_INSIDE_WEIGHTS and the float64
BBOX_NORMALIZE_MEANS / _STDS travel to the kernel as they are in cfg."""
import numpy as np
import numpy.random as npr
import torch

from frcnn_hip import ops
from model.config import cfg


def proposal_target_layer(rpn_rois, rpn_scores, gt_boxes, _num_classes):
    all_rois = np.ascontiguousarray(rpn_rois, dtype=np.float32)
    all_scores = np.ascontiguousarray(rpn_scores, dtype=np.float32)
    gt_boxes = np.ascontiguousarray(gt_boxes, dtype=np.float32)
    t = cfg.TRAIN
    n_prop = all_rois.shape[0]
    cand = all_rois[:, 1:5]
    if t.USE_GT:                                                                                        # :30-36: the gt boxes join the candidates
        cand = np.vstack((cand, gt_boxes[:, :4]))                                                       # (the device appends the same rows itself)
    num_images = 1
    rois_per_image = t.BATCH_SIZE / num_images                                                         # :39-40
    fg_rois_per_image = np.round(t.FG_FRACTION * rois_per_image)
    dev = torch.device("cuda", torch.cuda.current_device())
    rois_d = torch.from_numpy(all_rois).to(dev)
    gt_d = torch.from_numpy(gt_boxes).to(dev)
    cand_d = torch.from_numpy(np.ascontiguousarray(cand, dtype=np.float64)).to(dev)
    overlaps = ops.bbox_overlaps(cand_d, gt_d[:, :4].double().contiguous()).cpu().numpy()               # :104-106
    max_overlaps = overlaps.max(axis=1)
    fg_inds = np.where(max_overlaps >= t.FG_THRESH)[0]                                                  # :111
    bg_inds = np.where((max_overlaps < t.BG_THRESH_HI) & (max_overlaps >= t.BG_THRESH_LO))[0]          # :114-115
    if fg_inds.size > 0 and bg_inds.size > 0:                                                           # :118-135
        fg_rois_per_image = min(fg_rois_per_image, fg_inds.size)
        fg_inds = npr.choice(fg_inds, size=int(fg_rois_per_image), replace=False)
        bg_rois_per_image = rois_per_image - fg_rois_per_image
        to_replace = bg_inds.size < bg_rois_per_image
        bg_inds = npr.choice(bg_inds, size=int(bg_rois_per_image), replace=to_replace)
    elif fg_inds.size > 0:
        to_replace = fg_inds.size < rois_per_image
        fg_inds = npr.choice(fg_inds, size=int(rois_per_image), replace=to_replace)
        fg_rois_per_image = rois_per_image
    elif bg_inds.size > 0:
        to_replace = bg_inds.size < rois_per_image
        bg_inds = npr.choice(bg_inds, size=int(rois_per_image), replace=to_replace)
        fg_rois_per_image = 0
    else:
        raise ValueError("proposal_target_layer: no foreground and no background RoIs (the reference stops in pdb here, :133-135)")
    keep_inds = np.append(fg_inds, bg_inds).astype(np.int32)                                            # :138
    out = ops.proposal_target_layer_inject(rois_d, torch.from_numpy(all_scores.reshape(-1)).to(dev), gt_d, int(_num_classes),
                                           torch.from_numpy(keep_inds).to(dev), int(fg_rois_per_image),
                                           means=t.BBOX_NORMALIZE_MEANS, stds=t.BBOX_NORMALIZE_STDS,
                                           opts=ops.roi_target_opts(t.USE_GT, t.BBOX_INSIDE_WEIGHTS))
    rois, roi_scores, labels, bbox_targets, bbox_inside_weights, bbox_outside_weights = (o.cpu().numpy() for o in out)
    C = int(_num_classes)                                                                               # reshapes of :47-53
    return (rois.reshape(-1, 5), roi_scores.reshape(-1), labels.reshape(-1, 1), bbox_targets.reshape(-1, C * 4),
            bbox_inside_weights.reshape(-1, C * 4), bbox_outside_weights.reshape(-1, C * 4))
