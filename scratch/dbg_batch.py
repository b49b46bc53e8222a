import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "oracle"), os.path.join(R, "tf-faster-rcnn_amd"), os.path.join(R, "tf-faster-rcnn_amd", "lib")]
import numpy as np, torch
from frcnn_hip.runtime import Session
from model.config import cfg
from nets.resnet_v1 import resnetv1
cfg.TEST.RPN_POST_NMS_TOP_N = 48
sess = Session(seed=3); net = resnetv1(50)
net.create_architecture("TEST", 21, tag="default", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
sess.init_variables(net.variable_specs())
rng = np.random.RandomState(5)
image = (rng.rand(1, 150, 200, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
rng = np.random.RandomState(9)
img2 = (rng.rand(1, 150, 200, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)
im_info = np.array([150, 200, 1.0], dtype=np.float32)
keys = ("rois", "cls_score", "bbox_pred", "rpn_cls_prob", "rpn_bbox_pred", "rpn_cls_score")
singles = []
for im in (image, img2):
    p = net.forward_device(sess, net._stage_image(sess, im), im_info); torch.cuda.synchronize()
    d = {k: p[k].cpu().numpy().copy() for k in keys}; d["head"] = net._layers["head"].cpu().numpy().copy(); d["num"] = int(net._num_rois.item()); singles.append(d)
batch = net._stage_image(sess, np.concatenate([image, img2, image], axis=0))
p = net.forward_device(sess, batch, im_info); torch.cuda.synchronize()
per = net._rois_per_image
head = net._layers["head"].cpu().numpy()
print("num rois batch", net._num_rois.cpu().numpy(), "singles", [s["num"] for s in singles])
rel = lambda a, b: float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
for b, want in enumerate((singles[0], singles[1], singles[0])):
    sl = slice(b * per, (b + 1) * per)
    print(b, "head", rel(head[b:b+1], want["head"]), "rpn_score", rel(p["rpn_cls_score"][b:b+1].cpu().numpy(), want["rpn_cls_score"]),
          "rpn_prob", rel(p["rpn_cls_prob"][b:b+1].cpu().numpy(), want["rpn_cls_prob"]), "rpn_box", rel(p["rpn_bbox_pred"][b:b+1].cpu().numpy(), want["rpn_bbox_pred"]),
          "rois", float(np.abs(p["rois"][sl].cpu().numpy() - want["rois"]).max()), "cls_score", rel(p["cls_score"][sl].cpu().numpy(), want["cls_score"]))
