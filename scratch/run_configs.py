"""Runs every BASELINE.json config once at full size on the device chain (shape robustness + timing)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "tf-faster-rcnn_amd"), os.path.join(R, "tf-faster-rcnn_amd", "lib")]
import numpy as np, torch
from frcnn_hip.runtime import Session
from model.config import cfg
from nets.resnet_v1 import resnetv1
from nets.vgg16 import vgg16
from nets.mobilenet_v1 import mobilenetv1

def run_infer(name, net, H, W, scales, classes, post, steps=6, gain=1.0):
    cfg.TEST.RPN_POST_NMS_TOP_N = post
    sess = Session(seed=3)
    net.create_architecture("TEST", classes, tag=name, anchor_scales=scales, anchor_ratios=(0.5, 1, 2))
    if hasattr(net, "_fused_tail_entry"): net._fuse_tail_entry = True
    sess.init_variables(net.variable_specs())
    rng = np.random.RandomState(3)
    img = ((rng.rand(1, H, W, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(gain)
    im_info = np.array([H, W, 1.6], dtype=np.float32)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        d = net._stage_image(sess, img)
        for _ in range(2): out, cnt = net.detect_device(sess, d, im_info, (int(H / 1.6), int(W / 1.6)))
        st.synchronize(); t0 = time.time()
        for _ in range(steps): out, cnt = net.detect_device(sess, d, im_info, (int(H / 1.6), int(W / 1.6)))
        st.synchronize(); dt = (time.time() - t0) / steps
    print("%-44s %7.2f ms/img  %6.1f img/s  rois=%d dets=%d  GFLOP=%.1f" % (name, dt * 1e3, 1 / dt, int(net._num_rois.item()), int(cnt.item()), sess.flops_last_forward / 1e9), flush=True)
    cfg.TEST.RPN_POST_NMS_TOP_N = 300

run_infer("C1 VGG16 VOC 600x1000 (single image)", vgg16(), 600, 1000, (8, 16, 32), 21, 300, gain=1 / 64.0)
run_infer("C2 ResNet-101 VOC 600x1000 (single image)", resnetv1(101), 600, 1000, (8, 16, 32), 21, 300)
run_infer("C3 ResNet-101 COCO 800x1333 A=15 R=1000", resnetv1(101), 800, 1333, (2, 4, 8, 16, 32), 81, 1000)
run_infer("C4 MobileNet-v1 COCO 600x1000 A=12", mobilenetv1(), 600, 1000, (4, 8, 16, 32), 81, 300)
# C5: ResNet-152 COCO train step
from model.train_val import SolverWrapper, synthetic_data_layer
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
sess = Session(seed=3); net = resnetv1(152)
net.create_architecture("TRAIN", 81, tag="c5", anchor_scales=(4, 8, 16, 32), anchor_ratios=(0.5, 1, 2))
sess.init_variables(net.variable_specs())
sw = SolverWrapper(sess, net, synthetic_data_layer(81, seed=3, image_gain=1 / 256.0))
sw.train_model(2, verbose=False); torch.cuda.synchronize(); t0 = time.time()
h = sw.train_model(5, verbose=False); torch.cuda.synchronize()
print("%-44s %7.2f ms/step  (total loss %.3f -> %.3f)" % ("C5 ResNet-152 COCO train step 600x1000 R=256", (time.time() - t0) / 5 * 1e3, h[0], h[-1]))
