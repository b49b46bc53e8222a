"""GPU, world_size 2 (two processes on the one GPU of the test box, gloo backend -- RCCL refuses two ranks on one device):
the data-parallel TRAINING path end to end: two replicas with the same initial weights and different
images run SolverWrapper steps with the bucketed, overlapped gradient all-reduce (frcnn_hip.parallel.BucketedAllReduce driven
from TrainState.backward / apply); after every step both replicas must hold bit-identical weights and momentum, and the
weights must have moved."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "tf-faster-rcnn_amd"), os.path.join(root, "tf-faster-rcnn_amd", "lib")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from frcnn_hip import parallel
        from frcnn_hip.runtime import Session
        from model.config import cfg
        from model.train_val import SolverWrapper, synthetic_data_layer
        from nets.resnet_v1 import resnetv1
        cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.BG_THRESH_LO, cfg.TRAIN.DOUBLE_BIAS = 64, 0.0, False
        cfg.TRAIN.RPN_POST_NMS_TOP_N, cfg.TRAIN.RPN_PRE_NMS_TOP_N = 300, 2000
        sess = Session(seed=3)                                              # same weights on both ranks
        net = resnetv1(num_layers=50)
        net.create_architecture("TRAIN", 21, tag="dp", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess.init_variables(net.variable_specs())
        data = synthetic_data_layer(21, seed=100 + rank, height=160, width=208, scale=1.0, image_gain=1 / 64.0)   # different images
        ar = parallel.make_grad_all_reduce(bucket_bytes=8 << 20)
        sw = SolverWrapper(sess, net, data, all_reduce=ar, world_size=world, write_snapshots=False)
        digests = []
        for _ in range(2):
            sw.train_model(1, verbose=False)
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for sc in sorted(sw.state.params):
                p = sw.state.params[sc]
                h.update(p.w.cpu().numpy().tobytes())
                h.update(p.acc_w.cpu().numpy().tobytes())
            digests.append(h.hexdigest())
        first = sorted(sw.state.params)[0]
        moved = float((sw.state.params[first].acc_w.abs().sum()).item())
        q.put((rank, digests, moved, len(sw.state.params), int(sw.state.flat.numel())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_replicas_stay_identical_through_overlapped_all_reduce(dev):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (r0, d0, m0, n0, f0), (r1, d1, m1, n1, f1) = res
    assert (r0, r1) == (0, 1) and n0 == n1 > 40 and f0 == f1 > 20_000_000
    assert d0 == d1, "replicas diverged"                 # same weights AND momentum after each of the two steps
    assert d0[0] != d0[1] and m0 > 0                      # and they did train
