"""Host share of one C5 training step: enqueue time of ONE step into an empty queue vs. its completion time."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
side = int(os.environ.get("SIDE", "1"))
cfg.HIP.WGRAD_STREAM = bool(side)
c = b.CONFIGS["c5"]; dev = torch.device("cuda:0")
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
sess = Session(device=dev, seed=cfg.RNG_SEED)
net = b.make_net(c)
net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=b.ANCHOR_RATIOS)
sess.init_variables(net.variable_specs())
sw = SolverWrapper(sess, net, b.resident_blobs(synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED, image_gain=1 / 256.0), dev))
sw.train_model(5, verbose=False); torch.cuda.synchronize()
hs, ts = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sw.train_model(1, verbose=False); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(1e3 * (t1 - t0)); ts.append(1e3 * (t2 - t0))
print("side", side, "host enqueue ms/step", np.round(hs, 2), "complete ms", np.round(ts, 2))
