import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "oracle"), os.path.join(R, "tf-faster-rcnn_amd"), os.path.join(R, "tf-faster-rcnn_amd", "lib")]
import numpy as np, torch
import frcnn_oracle as ora
from frcnn_hip.runtime import Session
from model.config import cfg
from nets.resnet_v1 import resnetv1
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO = 64, 0.0
sess = Session(seed=5)
net = resnetv1(num_layers=50)
net.create_architecture("TRAIN", 21, tag="train", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
sess.init_variables(net.variable_specs())
rng = np.random.RandomState(2)
H, W = 128, 160
image = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)
blobs = dict(data=image, im_info=np.array([H, W, 1.0], dtype=np.float32), gt_boxes=gt)
net.train_forward(sess, blobs)
print("counts", net._proposal_targets["counts"].cpu().numpy())
rois = net._predictions["rois"].cpu().numpy()
print("rois head", rois[:5], "nonzero rows", int((np.abs(rois).sum(1) > 0).sum()))
ov = ora.bbox_overlaps(rois[:, 1:5], gt[:, :4])
print("max ov per gt", ov.max(axis=0), "n>=0.5", int((ov.max(1) >= 0.5).sum()))
bp = net._predictions["rpn_bbox_pred"].cpu().numpy(); print("bbox_pred absmax", np.abs(bp).max(), "cls prob range", net._predictions["rpn_cls_prob"].min().item(), net._predictions["rpn_cls_prob"].max().item())
lab = net._anchor_targets["rpn_labels"].cpu().numpy(); print("anchor labels fg/bg", (lab==1).sum(), (lab==0).sum())
