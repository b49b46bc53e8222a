"""layer_utils.anchor_target_layer -- the tf.py_func seam of lib/nets/network.py:164-168 with the signature and outputs
of lib/layer_utils/anchor_target_layer.py:18-138, computed by libfrcnn_hip.so.

"Host-oracle" sampling: the reference subsamples fg / bg anchors with two `npr.choice`
calls on numpy's GLOBAL MT19937 stream (:72-86).  This mirror makes exactly those calls -- same candidate arrays, sizes
and order -- so a run seeded like the reference (`np.random.seed(cfg.RNG_SEED)`, tools/trainval_net.py) consumes the
stream identically and returns the reference's arrays bit for bit.  IoU, argmax assignments, labels, regression targets,
weights and output layouts are device work (frcnn_anchor_target_layer / _inject); the host only draws the indices.  The
in-graph training step (nets/network.py) uses the device-side counter-hash sampler instead and never comes here."""
import numpy as np
import numpy.random as npr
import torch

from frcnn_hip import ops
from model.config import cfg


def anchor_target_layer(rpn_cls_score, gt_boxes, im_info, _feat_stride, all_anchors, num_anchors):
    A = int(num_anchors)
    height, width = rpn_cls_score.shape[1:3]
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    dev = torch.device("cuda", torch.cuda.current_device())
    all_anchors = np.asarray(all_anchors)
    gt = torch.from_numpy(np.ascontiguousarray(gt_boxes, dtype=np.float32)).to(dev)
    base_d = torch.from_numpy(np.ascontiguousarray(all_anchors[:A], dtype=np.float64)).to(dev)      # shift (0,0): the base anchors
    t = cfg.TRAIN
    args = (gt, float(im_info[0]), float(im_info[1]), int(height), int(width), base_d)
    kw = dict(feat_stride=stride, rpn_batchsize=int(t.RPN_BATCHSIZE), fg_fraction=float(t.RPN_FG_FRACTION),
              pos_overlap=float(t.RPN_POSITIVE_OVERLAP), neg_overlap=float(t.RPN_NEGATIVE_OVERLAP),
              opts=ops.rpn_target_opts(t.RPN_CLOBBER_POSITIVES, t.RPN_POSITIVE_WEIGHT, t.RPN_BBOX_INSIDE_WEIGHTS))     # :57-70, :91-109
    # labels before subsampling, in the reference's output layout (1,1,A*H,W) -> per-anchor order (h, w, a)
    pre = ops.anchor_target_layer(*args, seed=-1, **kw)[0].cpu().numpy()
    labels_all = pre.reshape(1, A, height, width).transpose(0, 2, 3, 1).reshape(-1)
    inds_inside = np.where((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) &
                           (all_anchors[:, 2] < im_info[1]) & (all_anchors[:, 3] < im_info[0]))[0]       # :31-36
    labels = labels_all[inds_inside]
    disable = []
    num_fg = int(t.RPN_FG_FRACTION * t.RPN_BATCHSIZE)                                                     # :72-78
    fg_inds = np.where(labels == 1)[0]
    if len(fg_inds) > num_fg:
        d = npr.choice(fg_inds, size=(len(fg_inds) - num_fg), replace=False)
        labels[d] = -1
        disable.append(d)
    num_bg = t.RPN_BATCHSIZE - np.sum(labels == 1)                                                       # :80-86
    bg_inds = np.where(labels == 0)[0]
    if len(bg_inds) > num_bg:
        d = npr.choice(bg_inds, size=(len(bg_inds) - num_bg), replace=False)
        disable.append(d)
    dis = inds_inside[np.concatenate(disable)].astype(np.int32) if disable else np.zeros((0,), dtype=np.int32)
    out = ops.anchor_target_layer_inject(*args, torch.from_numpy(dis).to(dev), **kw)
    return tuple(o.cpu().numpy() for o in out)
