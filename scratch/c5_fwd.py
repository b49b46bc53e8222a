"""Forward-only (TRAIN mode) pass of configs[4] for a kernel profile."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
c = b.CONFIGS["c5"]; dev = torch.device("cuda:0")
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
sess = Session(device=dev, seed=cfg.RNG_SEED)
net = b.make_net(c)
net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=b.ANCHOR_RATIOS)
sess.init_variables(net.variable_specs())
layer = b.resident_blobs(synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED, image_gain=1 / 256.0), dev)
sw = SolverWrapper(sess, net, layer)
sw.train_model(3, verbose=False); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    net.train_forward(sess, next(layer)); net._sample_seed += 2
torch.cuda.synchronize()
print("forward ms", (time.perf_counter() - t0) * 100)
