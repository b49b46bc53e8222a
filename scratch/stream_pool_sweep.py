"""Next-round tool (written at the end of round 4, not yet run): the ResNet-152 training step as a function of WHICH torch pool streams
the reverse sweep's helpers get.  Round 4 found (profiles/r04_ao_dp_pieces.txt, r04_ap_hw_queues.txt, r04_aq_side_stream_probe.txt) that
the step time depends on the assignment of streams to the 4 hardware queues: an RCCL group that merely exists costs 4 ms, taking 8 more
pool streams before the side streams moves the data-parallel-rules step from 23.7 to 19.8 ms and the single-GPU step from 17.3 to 19.8.
This script measures the step for `skip` = 0 .. 11 pool streams burnt before the sweep's side / solver / filter-preparation streams are
created, with and without a one-rank RCCL group, so that the assignment can be chosen by measurement (and then made deliberate in
frcnn_hip/train.py: e.g. a start-up autotune over a few skips, 2 steps each).

    python scratch/stream_pool_sweep.py [out.txt [skip,skip,...]]        # ~12 s per line on an MI355X
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import torch
import bench
from frcnn_hip.runtime import Session
from model.config import cfg
from model.train_val import SolverWrapper, synthetic_data_layer
skip, group = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if group:
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() %% 2000))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 256, 0.0, False
c = bench.CONFIGS["c5"]
sess = Session(device=dev, seed=cfg.RNG_SEED)
burnt = [torch.cuda.Stream(device=dev) for _ in range(skip)]            # pool streams nobody uses: they only shift the round-robin
net = bench.make_net(c)
net.create_architecture("TRAIN", c["classes"], tag="c5", anchor_scales=c["scales"], anchor_ratios=bench.ANCHOR_RATIOS)
sess.init_variables(net.variable_specs())
ar = None
if group:
    from frcnn_hip import parallel
    ar = parallel.make_grad_all_reduce()
sw = SolverWrapper(sess, net, bench.resident_blobs(synthetic_data_layer(c["classes"], seed=3, image_gain=1 / 256.0), dev), all_reduce=ar,
                   force_dp=bool(group))
sw.train_model(5, verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
sw.train_model(20, verbose=False)
torch.cuda.synchronize()
print("RESULT skip %%d group %%d: %%.2f ms per step" %% (skip, group, 1e3 * (time.perf_counter() - t0) / 20))
'''


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    skips = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(12))
    for group in (0, 1):
        for skip in skips:
            t0 = time.time()
            p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(skip), str(group)], capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            out.write((line[-1] if line else "skip %d group %d: FAILED %s" % (skip, group, p.stderr[-300:])) + "   (%.0f s)\n" % (time.time() - t0))
            out.flush()


if __name__ == "__main__":
    main()
