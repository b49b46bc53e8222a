"""Data-parallel inference: one process per GPU, one image per rank, and ONE collective per batch --
an all-gather of a fixed-size padded detection record (new relative to the reference,
which is strictly single-GPU / batch-1: lib/model/test.py:88).

Record layout (float32, REC_FLOATS long): rows [0, REC_ROWS) = (x1,y1,x2,y2,score,class) padded with
zeros, then the detection count as a float32 (exact below 2**24), then (rank, step) stamps (set_stamp, optional) and padding to
a 32-byte multiple.
The same code runs over RCCL/xGMI (backend "nccl", GPU tensors) and over gloo (CPU tensors, tests).
"""
import torch

REC_ROWS = 128
REC_FLOATS = REC_ROWS * 6 + 8


def new_record(device, batch=None):
    """-> (record, detection view).  batch=None: one image ([REC_FLOATS], [REC_ROWS,6]); batch=B: B images
    ([B,REC_FLOATS], [B,REC_ROWS,6] strided view whose per-image slices are contiguous)."""
    if batch is None:
        rec = torch.zeros((REC_FLOATS,), dtype=torch.float32, device=device)
        return rec, rec[:REC_ROWS * 6].view(REC_ROWS, 6)
    rec = torch.zeros((batch, REC_FLOATS), dtype=torch.float32, device=device)
    return rec, rec[:, :REC_ROWS * 6].unflatten(1, (REC_ROWS, 6))


def set_count(rec, count_i32):
    """count_i32: int32 tensor [1] (or [B] for a batched record) written by frcnn_detect_post."""
    if rec.dim() == 1:
        rec[REC_ROWS * 6] = count_i32[0].to(torch.float32)
    else:
        rec[:, REC_ROWS * 6] = count_i32.to(torch.float32)


def set_stamp(rec, rank, step):
    """(rank, step mod 2^20) into two of the record's padding floats (exact in float32): lets a receiver verify whose record of which
    step occupies a slot of the all-gather (bench.py's self-check)."""
    v = torch.tensor([float(rank), float(int(step) & 0xfffff)], dtype=torch.float32, device=rec.device)
    if rec.dim() == 1:
        rec[REC_ROWS * 6 + 1:REC_ROWS * 6 + 3] = v
    else:
        rec[:, REC_ROWS * 6 + 1:REC_ROWS * 6 + 3] = v


def get_stamp(rec_row):
    return int(rec_row[REC_ROWS * 6 + 1].item()), int(rec_row[REC_ROWS * 6 + 2].item())


def shard_images(num_images, rank, world):
    """Image i goes to rank i % world (weights replicated, nothing couples two images)."""
    return list(range(rank, num_images, world))


def all_gather_records(rec, gathered=None, group=None):
    """-> tensor [world, REC_FLOATS]; one all_gather_into_tensor call (latency bound, 3 KB/rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(rec.shape), dtype=torch.float32, device=rec.device)
    if dist.get_backend(group) == "gloo":
        parts = list(gathered.unbind(0))
        dist.all_gather(parts, rec, group=group)
    else:
        dist.all_gather_into_tensor(gathered, rec, group=group)
    return gathered


def unpack_records(gathered):
    """-> list over ranks of float32 [n_i, 6] detection arrays (host)."""
    g = gathered.detach().cpu().reshape(-1, REC_FLOATS)      # [world(*batch), REC_FLOATS]
    out = []
    for r in range(g.shape[0]):
        n = min(int(g[r, REC_ROWS * 6].item()), REC_ROWS)
        out.append(g[r, :REC_ROWS * 6].view(REC_ROWS, 6)[:n].clone().numpy())
    return out


def make_grad_all_reduce(group=None, bucket_bytes=64 << 20, overlap=True):
    """overlap=True (default): a BucketedAllReduce -- buckets leave as soon as their gradients are final during the reverse
    sweep.  overlap=False: the round-1 form below, all buckets after the sweep.

    Data-parallel training exchange: returns f(flat) that sums the flat f32 gradient buffer
    over the ranks in place, in contiguous buckets (default 64 MiB: large enough that the xGMI ring is
    bandwidth- not latency-bound, small enough that a later round can overlap them with the remaining
    backward sweep).  The mean over replicas is applied inside the SGD kernel (grad_scale = 1/world)."""
    import torch.distributed as dist
    if overlap:
        return BucketedAllReduce(group, bucket_bytes)

    def all_reduce(flat):
        n = flat.numel()
        step = max(1, bucket_bytes // 4)
        handles = [dist.all_reduce(flat[o:min(n, o + step)], op=dist.ReduceOp.SUM, group=group, async_op=True)
                   for o in range(0, n, step)]
        for h in handles:
            h.wait()
        return flat
    return all_reduce


class BucketedAllReduce(object):
    """Gradient exchange overlapped with the reverse sweep ("bucketed all-reduce ... in backward order,
    overlapped with the remaining backward").

    The flat gradient buffer is laid out in FORWARD order and the reverse sweep finishes it from the END: `ready(flat,
    data_ptr)` says that everything from the parameter at `data_ptr` to the end of the buffer is final; every time at least
    `bucket_bytes` of not-yet-sent gradient is final, one asynchronous `all_reduce(SUM)` over that contiguous range is issued
    (torch.distributed orders it after the kernels already enqueued on the current stream and runs it on the communicator's
    own stream, so the remaining backward kernels overlap it).  `finish(flat)` sends what is left (the front of the buffer)
    and waits for every range.  INVARIANT (asserted): the offsets passed to ready() never increase, i.e. the flat buffer is laid
    out in the reverse of the backward visiting order and every parameter has exactly ONE tape record (a filter shared by two
    records would be final only after its second visit; TrainState.build refuses such graphs).  xGMI is point-to-point: 64 MiB buckets keep the
    ring bandwidth-bound (a ResNet-152 step exchanges 253 MB)."""

    def __init__(self, group=None, bucket_bytes=64 << 20):
        self.group, self.bucket = group, max(1, int(bucket_bytes) // 4)
        self.reset()

    host_s = 0.0                       # host seconds spent inside torch.distributed calls (sends + waits), cumulative: bench.py reports it
    sends = 0

    def reset(self):
        self.sent_from = None          # element index: [sent_from, numel) is already in flight / done
        self.final_from = None
        self.handles = []

    multi_stream = True                # ready() takes every stream the gradients may have been enqueued on (TrainState._sweep: two side streams)

    def _order_after(self, streams):
        """One stream that is behind everything enqueued so far on `streams`: the collective is issued with it current (torch.distributed
        orders a collective after torch's current stream only)."""
        from . import ops
        streams = [st for st in (streams or []) if st is not None]
        if len(streams) <= 1:
            return streams[0] if streams else None
        if getattr(self, "_order", None) is None:
            self._order = torch.cuda.Stream(device=streams[0].device)
        for st in streams:
            ops.st_wait_stream(self._order, st)
        return self._order

    def _send(self, flat, lo, hi, stream=None):
        """One asynchronous all-reduce over flat[lo:hi].  A step that is being recorded (frcnn_hip/replay.py) keeps the call as a host
        operation of the step: a replayed step issues the same collectives at the same points of its launch list."""
        import time
        import torch.distributed as dist
        from . import ops
        import frcnn_hip as _binding
        if hi <= lo:
            return
        R = _binding.recorder
        slot = None if (stream is None or R is None) else R.slot(stream)

        def send(rec):
            t0 = time.perf_counter()
            self.sends += 1
            st = stream if (rec is None or slot is None) else rec.bound[slot]
            if st is None:
                self.handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:                      # the collective orders itself after torch's CURRENT stream: make `stream` current for this call only
                with torch.cuda.stream(st):
                    self.handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.host_s += time.perf_counter() - t0
        ops.host_op(send)

    def wait_sent(self, stream):
        """`stream` waits for every all-reduce issued so far (the in-sweep solver: an update reads the summed gradients).  RCCL: a
        stream-level wait, the host does not block; gloo (tests): the host waits."""
        from . import ops
        import frcnn_hip as _binding
        R = _binding.recorder
        slot = None if R is None else R.slot(stream)

        def wait(rec):
            st = stream if (rec is None or slot is None) else rec.bound[slot]
            with torch.cuda.stream(st):
                for h in self.handles:
                    h.wait()
        ops.host_op(wait)

    def ready(self, flat, data_ptr, stream=None, streams=None):
        """stream: the stream this parameter's gradient kernels were enqueued on; streams: every stream gradients of LATER parameters
        (already visited by the reverse sweep) may still be running on -- the collective is ordered after all of them.  Only a call
        that actually sends a bucket touches streams -- a handful per step, not one per parameter (the per-parameter
        `with torch.cuda.stream(...)` of round 3 cost the host ~4 ms of a 20 ms step)."""
        n = flat.numel()
        if self.sent_from is None:
            self.sent_from = n
        off = (int(data_ptr) - flat.data_ptr()) // 4
        if off < 0 or off > n:
            return
        if self.final_from is not None and off > self.final_from:
            raise RuntimeError("BucketedAllReduce.ready: offsets must not increase (%d after %d): the flat gradient buffer is not in "
                               "reverse backward order, or a parameter has more than one tape record" % (off, self.final_from))
        self.final_from = off
        order = None
        while self.sent_from - self.final_from >= self.bucket:
            if order is None:
                order = self._order_after(streams if streams else [stream])
            lo = self.sent_from - self.bucket
            self._send(flat, lo, self.sent_from, order)
            self.sent_from = lo

    def finish(self, flat):
        n = flat.numel()
        hi = n if self.sent_from is None else self.sent_from
        step = self.bucket
        while hi > 0:                                    # the rest, still back to front, in bucket-sized pieces
            lo = max(0, hi - step)
            self._send(flat, lo, hi)
            hi = lo
        from . import ops
        ops.host_op(self._wait_all)
        return flat

    def _wait_all(self):
        import time
        t0 = time.perf_counter()
        for h in self.handles:
            h.wait()
        self.host_s += time.perf_counter() - t0
        self.reset()

    def __call__(self, flat):                            # plain callable form (no overlap)
        return self.finish(flat)
