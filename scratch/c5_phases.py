"""C5 step by phase: host enqueue time vs GPU completion time of forward / backward / solver, each into an empty queue."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from frcnn_hip.runtime import Session
from frcnn_hip import ops
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
cfg.HIP.WGRAD_STREAM = int(os.environ.get("SIDE", "2"))
cfg.HIP.WGRAD_TN = bool(int(os.environ.get("TN", "1")))
c = b.CONFIGS["c5"]; dev = torch.device("cuda:0")
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
sess = Session(device=dev, seed=cfg.RNG_SEED)
net = b.make_net(c)
net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=b.ANCHOR_RATIOS)
sess.init_variables(net.variable_specs())
layer = b.resident_blobs(synthetic_data_layer(c["classes"], seed=cfg.RNG_SEED, image_gain=1 / 256.0), dev)
sw = SolverWrapper(sess, net, layer)
sw.train_model(5, verbose=False); torch.cuda.synchronize()
ts = sw.state
if os.environ.get("NOWGRAD"):
    ops.conv2d_wgrad = lambda *a, **k: None
    ops.colsum = lambda *a, **k: None
n_launch = [0]
orig_call = ops.call
rows = []
for _ in range(6):
    blobs = next(layer)
    r = []
    def phase(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        r.extend([1e3 * (t1 - t0), 1e3 * (t2 - t0)])
    phase(lambda: net.train_forward(sess, blobs))
    phase(lambda: ts.backward(net._loss_seeds))
    def solver():
        total = torch.empty((1,), dtype=torch.float32, device=dev)
        ts.regularization_loss(total)
        ts.apply(ts.lr, 1, None)
    phase(solver)
    net._sample_seed += 2
    rows.append(r)
a = np.array(rows)[1:].mean(axis=0)
print("side streams", cfg.HIP.WGRAD_STREAM, "wgrad_tn", cfg.HIP.WGRAD_TN)
print("forward : host %.2f ms, done %.2f ms" % (a[0], a[1]))
print("backward: host %.2f ms, done %.2f ms" % (a[2], a[3]))
print("solver  : host %.2f ms, done %.2f ms" % (a[4], a[5]))
