import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "tf-faster-rcnn_amd"), os.path.join(R, "tf-faster-rcnn_amd", "lib")]
import numpy as np, torch
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
from nets.resnet_v1 import resnetv1
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS, cfg.TRAIN.DISPLAY = 256, 0.0, False, 5
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 101
sess = Session(seed=3)
net = resnetv1(num_layers=layers)
net.create_architecture("TRAIN", 21, tag='default', anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2))
sess.init_variables(net.variable_specs())
data = synthetic_data_layer(21, seed=3, image_gain=1.0 / 256.0)
sw = SolverWrapper(sess, net, data)
sw.train_model(3, verbose=False)
torch.cuda.synchronize(); t0 = time.time()
h = sw.train_model(10, verbose=True)
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print("res%d train step: %.1f ms/iter (600x1000, 256 rois, 1 GPU), losses %s" % (layers, dt * 1e3, ["%.3f" % x for x in h[-3:]]))
