import sys, os, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "tf-faster-rcnn_amd"), os.path.join(R, "tf-faster-rcnn_amd", "lib")]
import numpy as np, torch
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper
from nets.resnet_v1 import resnetv1
dev = torch.device("cuda:0")
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.SNAPSHOT_ITERS, cfg.TRAIN.STEPSIZE = 64, 0.0, 2e-4, 2, [3]
cfg.TRAIN.DISPLAY = 1000
rng = np.random.RandomState(2)
image = ((rng.rand(1, 128, 160, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(1 / 256.0)
gt = np.array([[16, 16, 79, 79, 3], [60, 30, 150, 110, 7], [5, 70, 60, 120, 12]], dtype=np.float32)
def layer():
    while True:
        yield dict(data=image, im_info=np.array([128, 160, 1.0], dtype=np.float32), gt_boxes=gt)
def solver(tag):
    sess = Session(device=dev, seed=5); net = resnetv1(num_layers=50)
    net.create_architecture("TRAIN", 21, tag=tag, anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
    sess.init_variables(net.variable_specs()); return sess, net, SolverWrapper(sess, net, layer())
d = tempfile.mkdtemp()
_, _, a = solver("A"); fa = a.train_model(3, verbose=False); wa = a.state.export_variables(True)
_, _, b = solver("B"); fb = b.train_model(2, verbose=False, snapshot_dir=d); wb2 = b.state.export_variables(True)
_, _, c = solver("C"); c.restore(os.path.join(d, "res101_faster_rcnn_iter_2.ckpt"), os.path.join(d, "res101_faster_rcnn_iter_2.pkl"))
fc = c.train_model(3, verbose=False, start_iter=2); wc = c.state.export_variables(True)
print(fa, fb, fc)
# after import, were the slots equal to wb2's?  compare the step-3 result
worst = sorted(((float(np.abs(wa[k] - wc[k]).max() / max(np.abs(wa[k]).max(), 1e-12)), k) for k in wa), reverse=True)[:12]
for e, k in worst: print("%.3e %s" % (e, k))
