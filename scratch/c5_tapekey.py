"""Which tape tensors change address from step to step (they defeat backward_auto's graph key)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
cfg.HIP.TRAIN_GRAPH = False
c = b.CONFIGS["c5"]; dev = torch.device("cuda:0")
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
sess = Session(device=dev, seed=cfg.RNG_SEED)
net = b.make_net(c)
net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=b.ANCHOR_RATIOS)
sess.init_variables(net.variable_specs())
layer = b.resident_blobs(synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED, image_gain=1 / 256.0), dev)
sw = SolverWrapper(sess, net, layer)
prev = None
for step in range(6):
    sw.train_model(1, verbose=False)
    cur = {}
    for i, rec in enumerate(net._tape):
        for k, v in rec.items():
            if torch.is_tensor(v):
                cur[(i, rec["kind"], rec.get("scope", rec.get("name", "")), k)] = (v.data_ptr(), tuple(v.shape))
    for i, (t, g) in enumerate(net._loss_seeds):
        cur[("seed", i, "t")] = (t.data_ptr(), tuple(t.shape))
    if prev is not None:
        diff = [k for k in cur if prev.get(k) != cur[k]]
        print("step", step, "records", len(net._tape), "changed:", diff[:12], len(diff))
    prev = cur
