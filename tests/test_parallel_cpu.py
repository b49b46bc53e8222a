"""CPU, world_size 2, gloo: the N>1 path of bench.py -- image sharding + the single all-gather of
fixed-size detection records (frcnn_hip/parallel.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tf-faster-rcnn_amd"))
    from frcnn_hip import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = parallel.shard_images(5, rank, world)
        rng = np.random.RandomState(100 + rank)
        n = 3 + 4 * rank
        dets = rng.rand(n, 6).astype(np.float32)
        rec, view = parallel.new_record("cpu")
        view[:n] = torch.from_numpy(dets)
        parallel.set_count(rec, torch.tensor([n], dtype=torch.int32))
        out = parallel.unpack_records(parallel.all_gather_records(rec))
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # training exchange: bucketed gradient sum
        parallel.make_grad_all_reduce(bucket_bytes=1024, overlap=False)(flat)
        assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)
        # the overlapped form: ranges are handed over back to front while "the reverse sweep" is still running
        flat2 = torch.arange(5000, dtype=torch.float32) * (rank + 1)
        ar = parallel.make_grad_all_reduce(bucket_bytes=4096)                  # 1024-element buckets, BucketedAllReduce
        inflight = []
        for off in (4100, 2600, 1500, 700, 0):                                  # parameters in reverse forward order
            ar.ready(flat2, flat2[off:].data_ptr())
            inflight.append(len(ar.handles))
        assert inflight == [0, 2, 3, 4, 4], inflight                            # 900 final: nothing yet; 2400: two buckets; ...
        ar.finish(flat2)
        assert torch.equal(flat2, torch.arange(5000, dtype=torch.float32) * 3) and ar.handles == []
        q.put((rank, mine, [o.tolist() for o in out], dets.tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_all_gather_detection_records_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]          # image i -> rank i % world
    sent = [res[0][3], res[1][3]]
    for rank, _, gathered, _ in res:                                 # every rank holds every rank's detections
        assert len(gathered) == 2
        for r in range(2):
            assert np.array_equal(np.array(gathered[r], dtype=np.float32), np.array(sent[r], dtype=np.float32))


def _bench_worker(rank, world, port, q):
    """bench.py's OWN record path (exchange_records / run_timed / check_exchange) on CPU tensors under gloo: what `bench.py --gpus N`
    executes per step around the device work, with a stand-in for detect_device."""
    import importlib.util
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tf-faster-rcnn_amd")]
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    from frcnn_hip import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        B, S = 4, 3                                                     # the bench's default: 4 images per step, 3 chains
        recs, views, counts, gathered = [], [], [], []
        for i in range(S):
            r_, v_ = parallel.new_record("cpu", batch=B)
            recs.append(r_); views.append(v_)
            counts.append(torch.zeros((B,), dtype=torch.int32))
            gathered.append(torch.zeros((world,) + tuple(r_.shape), dtype=torch.float32))

        def fake_detect(i, k):                                          # deterministic "detections" of (rank, chain, step)
            n = 3 + (rank + i + k) % 5
            views[i].zero_()
            views[i][:, :n, :] = float(100 * rank + 10 * i + k % 7)
            counts[i][:] = n

        calls = []

        def step(k, stamp=None):
            i = k % S
            fake_detect(i, k)
            bench.exchange_records(recs[i], counts[i], gathered[i], rank, stamp)
            calls.append(k)
        t = bench.run_timed(step, 7, 3, dist)
        assert t > 0 and calls == [0, 1, 2] + list(range(7))            # W warm-up steps, then exactly K timed ones
        for i in range(S):
            step(i, stamp=1000 + i)
        ok = all(bench.check_exchange(recs[i], gathered[i], rank, world, 1000 + i) for i in range(S))
        # a stale slot is detected: checking against another step's stamp must fail
        try:
            bench.check_exchange(recs[0], gathered[0], rank, world, 999)
            stale_detected = False
        except RuntimeError:
            stale_detected = True
        other = 1 - rank
        peer_n = int(gathered[0][other, 0, parallel.REC_ROWS * 6].item())
        q.put((rank, ok, stale_detected, peer_n, float(gathered[0][other, 0, 0].item())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_record_path_world2_gloo():
    """bench.py's step / timed-region / all-gather record path, factored out of main(), runs for world 2 under gloo;
    the untimed self-check verifies every slot and rejects a stale one."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, stale, peer_n, peer_v in res:
        other = 1 - rank
        assert ok and stale
        assert peer_n == 3 + (other + 0 + 0) % 5 and peer_v == float(100 * other)       # chain 0's last step was the check step k = 0
