"""GPU, world_size 2 (two processes on the one GPU of the test box, gloo backend -- RCCL refuses two ranks on one device):
the data-parallel TRAINING path end to end: two replicas with the same initial weights and different
images run SolverWrapper steps with the bucketed, overlapped gradient all-reduce (frcnn_hip.parallel.BucketedAllReduce driven
from TrainState.backward / apply); after every step both replicas must hold bit-identical weights and momentum, and the
weights must have moved.  And the update must be the RIGHT one: the buffer the bucketed all-reduce leaves is g0 + g1 of the two replicas' own
gradients (each recomputed in a second, exchange-free session from the same weights / image / seeds), and the parameters after the step
equal a single-process update on their mean to <= 2 ulp.  Steps 3 and 4 are replayed from the recorded launch list (cfg.HIP.TRAIN_REPLAY):
the collectives are host operations of the list."""
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
        ts = sw.state
        ts.lr = cfg.TRAIN.LEARNING_RATE
        blob = next(data)

        def digest():
            h = hashlib.sha256()
            for sc in sorted(ts.params):
                p = ts.params[sc]
                h.update(p.w.cpu().numpy().tobytes())
                h.update(p.acc_w.cpu().numpy().tobytes())
            return h.hexdigest()

        # ---- step 1 under the data-parallel rules, and the SAME step rebuilt from its parts ------------------------------------------
        # (replicas that stay identical only prove that both applied the same thing; here the thing is checked: the all-reduced buffer is
        # g0 + g1 of the two replicas' own gradients, and the update equals a single-process update on their mean)
        net.train_step_async(sess, blob, ts)
        torch.cuda.synchronize()
        state0 = None
        after_dp = {sc: (p.w.clone(), p.acc_w.clone()) for sc, p in ts.params.items()}
        summed = ts.flat.clone()                                            # what the bucketed all-reduce left in the flat buffer
        d_first = digest()
        # rewind: the initial weights are a function of the session seed, momentum starts at zero
        sess2 = Session(seed=3)
        net2 = resnetv1(num_layers=50)
        net2.create_architecture("TRAIN", 21, tag="dp_ref", anchor_scales=(4, 8, 16), anchor_ratios=(0.5, 1, 2))
        sess2.init_variables(net2.variable_specs())
        from frcnn_hip.train import TrainState
        ts2 = TrainState(sess2, net2, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WEIGHT_DECAY, double_bias=cfg.TRAIN.DOUBLE_BIAS,
                         bias_decay=cfg.TRAIN.BIAS_DECAY)
        ts2.lr = ts.lr
        net2.train_forward(sess2, blob)
        ts2.build()
        net2.configure_train_op(ts2)
        ts2.backward(net2._loss_seeds)                                      # this replica's OWN gradient, no exchange, no update
        torch.cuda.synchronize()
        mine = ts2.flat.cpu()
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)                                         # (gloo, host tensors)
        want_sum = (both[0] + both[1]).to(summed.device)                    # two addends: the order cannot matter
        sum_ok = bool(torch.equal(summed, want_sum))
        sum_err = float((summed - want_sum).abs().max().item())
        ts2.flat.copy_(want_sum)
        ts2.apply(ts2.lr, world_size=world, all_reduce=None)                # grad_scale = 1 / world: the mean, in one process
        torch.cuda.synchronize()
        worst = 0.0
        for sc, p in ts2.params.items():
            for got, ref in zip(after_dp[sc], (p.w, p.acc_w)):
                worst = max(worst, float(((got - ref).abs() / (ref.abs() * 2.0 ** -23 + 1e-30)).max().item()))      # in ulps of the reference
        digests = [d_first]
        for _ in range(3):                                                  # steps 2-4: step 2 is recorded, 3 and 4 are REPLAYED -- collectives included
            net.train_step_async(sess, next(data), ts)
            torch.cuda.synchronize()
            digests.append(digest())
        first = sorted(ts.params)[0]
        moved = float((ts.params[first].acc_w.abs().sum()).item())
        q.put((rank, digests, moved, len(ts.params), int(ts.flat.numel()), sum_ok, sum_err, worst, dict(net.replay_stats)))
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
    (r0, d0, m0, n0, f0, s0, e0, w0, st0), (r1, d1, m1, n1, f1, s1, e1, w1, st1) = res
    assert (r0, r1) == (0, 1) and n0 == n1 > 40 and f0 == f1 > 20_000_000
    assert d0 == d1, "replicas diverged"                 # same weights AND momentum after each of the four steps
    assert len(set(d0)) == 4 and m0 > 0                   # and they did train
    assert s0 and s1, ("the all-reduced gradient buffer is not g0 + g1", e0, e1)
    assert max(w0, w1) <= 2.0, ("data-parallel update vs single-process update on the mean gradient, in ulps", w0, w1)
    assert st0 == st1 == dict(eager=1, recorded=1, replayed=2), (st0, st1)
