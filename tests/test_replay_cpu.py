"""CPU: the launch-list form of a training step (frcnn_hip/replay.py) against stand-ins for the C-ABI entries, streams and events:
a recording replays the same calls with the same arguments in the same order; streams are slots that can be bound to other physical
streams; the per-step arguments (sampling seeds, gt-box count) are rewritten at replay; a failing entry raises; the recording hooks in
frcnn_hip.call / frcnn_hip.ops append to the recorder only while one is installed; the arena hands out the same tensors every step."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd"))

import frcnn_hip                              # noqa: E402
from frcnn_hip import ops, replay             # noqa: E402


class FakeFn(object):
    """a C-ABI entry: logs (name, argument values) and returns a status code"""

    def __init__(self, name, argtypes, log, rc=0):
        self.__name__, self.argtypes, self.log, self.rc = name, argtypes, log, rc

    def __call__(self, *args):
        assert all(isinstance(a, t) for a, t in zip(args, self.argtypes)), "replayed arguments must arrive as the ctypes of the signature"
        self.log.append((self.__name__,) + tuple(a.value for a in args))
        return self.rc


class FakeStream(object):
    def __init__(self, name, handle, log):
        self.name, self.cuda_stream, self.log = name, handle, log

    def wait_event(self, ev):
        self.log.append(("wait", self.name, ev.name))


class FakeEvent(object):
    def __init__(self, name, log):
        self.name, self.log = name, log

    def record(self, stream):
        self.log.append(("record", self.name, stream.name))


P, I, LL, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_double


def record_a_step(log):
    main, side = FakeStream("main", 0x1000, log), FakeStream("side", 0x2000, log)
    ev = FakeEvent("e0", log)
    conv = FakeFn("frcnn_conv", [P, I, P], log)
    target = FakeFn("frcnn_target", [P, I, D, LL, P], log)
    plan = FakeFn("frcnn_conv2d_wgrad_set_plan", [I, I], log)
    rec = replay.Recording(main)
    rec.vars = dict(seed=10, gt=3)
    rec.slot(side)
    frcnn_hip.recorder = rec
    try:
        rec.add_call(conv, "frcnn_conv", (P(0xA0), 7, P(0x1000)))                 # a launch on main
        rec.add_call(target, "frcnn_target", (P(0xB0), 3, 0.5, 10 + 1, P(0x1000)))
        rec.patch_last(1, var="gt")
        rec.patch_last(3)
        ops.ev_record(ev, main)
        ops.st_wait_event(side, ev)
        rec.add_call(plan, "frcnn_conv2d_wgrad_set_plan", (0, 256))               # no stream argument although the last one is an int
        rec.add_call(conv, "frcnn_conv", (P(0xC0), 9, P(0x2000)))                 # a launch on the side stream
    finally:
        frcnn_hip.recorder = None
    return rec, main, side


def test_replay_repeats_the_recorded_step_and_rewrites_the_per_step_arguments():
    log = []
    rec, main, side = record_a_step(log)
    eager = list(log)
    assert eager == [("record", "e0", "main"), ("wait", "side", "e0")]            # (add_call does not execute; the ops helpers do)
    del log[:]
    rec.bind(rec.default_binding(main))
    rec.replay(dict(seed=10, gt=3))
    assert log == [("frcnn_conv", 0xA0, 7, 0x1000), ("frcnn_target", 0xB0, 3, 0.5, 11, 0x1000), ("record", "e0", "main"),
                   ("wait", "side", "e0"), ("frcnn_conv2d_wgrad_set_plan", 0, 256), ("frcnn_conv", 0xC0, 9, 0x2000)]
    del log[:]
    rec.replay(dict(seed=16, gt=40))                                              # three steps later, an image with 40 boxes
    assert log[1] == ("frcnn_target", 0xB0, 40, 0.5, 17, 0x1000)
    assert [e for i, e in enumerate(log) if i != 1] == [("frcnn_conv", 0xA0, 7, 0x1000), ("record", "e0", "main"), ("wait", "side", "e0"),
                                                        ("frcnn_conv2d_wgrad_set_plan", 0, 256), ("frcnn_conv", 0xC0, 9, 0x2000)]


def test_streams_are_slots_that_can_be_bound_to_other_streams():
    log = []
    rec, main, side = record_a_step(log)
    other_main, other_side = FakeStream("main2", 0x3000, log), FakeStream("side2", 0x4000, log)
    assert not rec.bound_to(other_main)
    rec.bind([other_main, other_side])
    assert rec.bound_to(other_main) and not rec.bound_to(main)
    del log[:]
    rec.replay(dict(seed=10, gt=3))
    assert log == [("frcnn_conv", 0xA0, 7, 0x3000), ("frcnn_target", 0xB0, 3, 0.5, 11, 0x3000), ("record", "e0", "main2"),
                   ("wait", "side2", "e0"), ("frcnn_conv2d_wgrad_set_plan", 0, 256), ("frcnn_conv", 0xC0, 9, 0x4000)]


def test_a_failing_entry_raises_at_replay():
    log = []
    main = FakeStream("main", 0x1000, log)
    rec = replay.Recording(main)
    rec.add_call(FakeFn("frcnn_bad", [P], log, rc=-2), "frcnn_bad", (P(0x1000),))
    rec.bind([main])
    with pytest.raises(frcnn_hip.FrcnnHipError, match="workspace too small"):
        rec.replay(dict(seed=0, gt=0))


def test_host_ops_run_now_and_at_every_replay_and_see_the_bound_streams():
    log = []
    main, side = FakeStream("main", 0x1000, log), FakeStream("side", 0x2000, log)
    rec = replay.Recording(main)
    frcnn_hip.recorder = rec
    try:
        slot = rec.slot(side)
        ops.host_op(lambda: log.append("plain"))
        ops.host_op(lambda r: log.append(("send on", side.name if r is None else r.bound[slot].name)))
    finally:
        frcnn_hip.recorder = None
    assert log == ["plain", ("send on", "side")]
    ops.host_op(lambda: log.append("not recorded"))                               # no recorder installed: runs, is not kept
    del log[:]
    rec.bind([main, FakeStream("side2", 0x5000, log)])
    rec.replay(dict(seed=0, gt=0))
    assert log == ["plain", ("send on", "side2")]


def test_tensor_operations_of_a_step_are_replayed_whatever_they_return():
    log = []
    main = FakeStream("main", 0x1000, log)
    a, b = torch.arange(6, dtype=torch.float32), torch.zeros(6)
    rec = replay.Recording(main)
    frcnn_hip.recorder = rec
    try:
        ops.t_copy(b, a)                       # (Tensor.copy_ / zero_ return the tensor: not a status code)
        ops.t_zero(a)
        ops.host_op(lambda: torch.add(b, 1.0, out=b))
    finally:
        frcnn_hip.recorder = None
    assert b.tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0] and float(a.abs().sum()) == 0.0
    a.copy_(torch.full((6,), 7.0))
    rec.bind([main])
    rec.replay(dict(seed=0, gt=0))
    assert b.tolist() == [8.0] * 6 and float(a.abs().sum()) == 0.0


def test_arena_hands_out_the_same_tensors_every_step_and_only_while_active():
    class Sess(object):
        buffers, device = {}, torch.device("cpu")
    a = replay.Arena(Sess(), "t")
    ops.arena = a
    try:
        x1, y1, z1 = ops._empty((2, 3)), ops._empty((2, 3)), ops._zeros((4,), dtype=torch.int32)
        a.reset()
        x2, y2, z2 = ops._empty((2, 3)), ops._empty((2, 3)), ops._zeros((4,), dtype=torch.int32)
    finally:
        ops.arena = None
    assert x1.data_ptr() == x2.data_ptr() and y1.data_ptr() == y2.data_ptr() and z1.data_ptr() == z2.data_ptr()
    assert x1.data_ptr() != y1.data_ptr() and z1.dtype == torch.int32 and int(z1.abs().sum()) == 0
    w = ops._empty((2, 3))
    assert w.data_ptr() not in (x1.data_ptr(), y1.data_ptr())                     # outside the arena: an ordinary allocation


def test_every_entry_without_a_stream_argument_is_classified():
    """A recording treats the LAST pointer argument of a launch as its stream.  Entries of include/frcnn_hip.h that do not end in
    `void* stream` must therefore be known: launch-context setters (recorded as they are), host-only functions (never recorded) or
    size queries (never called through frcnn_hip.call)."""
    import re
    h = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//.*", "", h)
    decls = re.findall(r"\b(?:int|size_t|void|unsigned int|long long|const char\*)\s+\*?\s*(_nms|frcnn_\w+)\s*\(([^;{]*)\)\s*;", h)
    assert len(decls) > 100
    without = {n for n, a in decls if not re.search(r"void\s*\*\s*stream\s*$", a.strip())}
    queries = {n for n in without if n.endswith("_bytes") or n.endswith("_supported")}
    assert without - queries <= (replay.NO_STREAM_ARG | replay.HOST_ONLY), sorted(without - queries - replay.NO_STREAM_ARG - replay.HOST_ONLY)
    assert not (replay.NO_STREAM_ARG & replay.HOST_ONLY)
    log = []
    rec = replay.Recording(FakeStream("main", 0x1000, log))
    rec.add_call(FakeFn("frcnn_generate_anchors", [I, P, I, P, I, P], log), "frcnn_generate_anchors", (16, P(1), 3, P(2), 3, P(3)))
    assert rec.cmds == []


# ---- the control flow of Network.train_step_async under cfg.HIP.TRAIN_REPLAY, with stand-ins for everything that needs a GPU -------------
class _Stream(FakeStream):
    def synchronize(self):
        self.log.append(("sync", self.name))

    def wait_stream(self, other):
        self.log.append(("wait_stream", self.name, other.name))


class _Prepared(object):
    def __init__(self):
        self.version, self.ready_version, self.plan, self.ready, self.readers, self.epoch = 3, 3, {"k": 1}, frozenset(["k"]), {}, 0

        self.gen, self.forgot = 0, []

    def refreshed(self):
        self.epoch += 1

    def forget_waits(self, stream):
        self.forgot.append(stream.name)


class _Sess(object):
    def __init__(self):
        self.graphs, self.buffers, self.device, self.prepared = {}, {}, torch.device("cpu"), _Prepared()
        self.derived_gen = 0

    def derived_generation(self):
        return (self.derived_gen, self.prepared.gen)

    def shape_scope(self, key, group=None, cap=None):
        import contextlib
        self.scope_log = getattr(self, "scope_log", []) + [(key, group, cap)]
        return contextlib.nullcontext()

    def buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        if key not in self.buffers:
            self.buffers[key] = torch.zeros(tuple(shape), dtype=dtype)
        return self.buffers[key]


class _TrainOp(object):
    lr, params, _sgd_table = 1e-3, {"w": 1}, object()

    def replay_signature(self):
        return ("sig",)


def _stub_net(monkeypatch, log):
    sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd", "lib"))
    from nets.network import Network
    main = _Stream("main", 0x1000, log)
    side = _Stream("side", 0x2000, log)
    made = []

    def new_stream(device=None):
        st = _Stream("pool%d" % len(made), 0x9000 + 16 * len(made), log)
        made.append(st)
        return st
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: main)
    monkeypatch.setattr(torch.cuda, "Stream", new_stream)
    conv = FakeFn("frcnn_conv", [P, I, P], log)
    target = FakeFn("frcnn_target", [P, I, LL, P], log)

    class Net(Network):
        def _train_scope(self, sess, blobs):
            return Network._train_scope(self, sess, dict(blobs, data=np.zeros(tuple(blobs["shape"][:3]) + (3,), dtype=np.float32)))

        def _stage_train_inputs(self, sess, blobs):
            self._sess, self._image, self._im_info = sess, torch.zeros(blobs["shape"]), (1.0, 2.0, 1.0)
            self._gt_boxes = sess.buf("gt", (128, 5))[:blobs["G"]]

        def _train_step_body(self, sess, train_op, out):
            if frcnn_hip.recorder is not None:
                frcnn_hip.recorder.slot(side)                   # (the real step meets a helper stream first in an event wait: ops.st_wait_event)
            for fn, name, args in ((conv, "frcnn_conv", (P(0xA0), 7, P(0x1000))), (target, "frcnn_target", (P(0xB0), int(self._gt_boxes.shape[0]), self._sample_seed, P(0x1000))),
                                   (conv, "frcnn_conv", (P(0xC0), 9, P(0x2000)))):
                fn(*[t(a) if not isinstance(a, ctypes._SimpleCData) else a for t, a in zip(fn.argtypes, args)])
                if frcnn_hip.recorder is not None:
                    frcnn_hip.recorder.add_call(fn, name, args)
                    if name == "frcnn_target":
                        frcnn_hip.recorder.patch_last(1, var="gt")
                        frcnn_hip.recorder.patch_last(2)
            self._predictions, self._losses, self._proposal_targets, self._anchor_targets = {"p": 1}, {"l": 1}, {"pt": 1}, {"at": 1}
            self._sample_seed += 2
            return out
    net = Net()
    net._mode, net._tag, net._num_classes, net._anchor_scales, net._anchor_ratios = "TRAIN", "t", 21, (8,), (1,)
    return net, main, side, made


def test_train_step_goes_eager_then_recorded_then_replayed(monkeypatch):
    from model.config import cfg
    log = []
    net, main, side, made = _stub_net(monkeypatch, log)
    sess, op = _Sess(), _TrainOp()
    old = (cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS)
    cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = True, 0
    try:
        for i, G in enumerate((3, 5, 2, 9)):
            del log[:]
            net.train_step_async(sess, dict(shape=(1, 4, 6, 4), G=G), op)
            assert [e for e in log if e[0] == "frcnn_target"] == [("frcnn_target", 0xB0, G, 2 * i, 0x1000)], (i, log)      # this step's box count and seed
            assert len([e for e in log if e[0] == "frcnn_conv"]) == 2
        assert net.replay_stats == dict(eager=1, recorded=1, replayed=2) and net._sample_seed == 8 and sess.prepared.epoch == 2
        # every step ran inside its image shape's buffer scope (Session.shape_scope), LRU group = the network tag
        assert sess.scope_log == [(("train_shape", "t", (1, 4, 6, 4)), ("train", "t"), int(cfg.HIP.TRAIN_CACHE_SHAPES))] * 4
        assert all(e.get("scope") == ("train_shape", "t", (1, 4, 6, 4)) for k, e in sess.graphs.items() if k[0] == "train_replay")
        net.train_step_async(sess, dict(shape=(1, 4, 8, 4), G=1), op)                 # another image shape: eager again
        assert net.replay_stats == dict(eager=2, recorded=1, replayed=2)
        sess.prepared.version += 1                                                    # filters changed behind the solver's back: not steady, no recording
        net.train_step_async(sess, dict(shape=(1, 4, 8, 4), G=1), op)
        assert net.replay_stats == dict(eager=3, recorded=1, replayed=2)
        cfg.HIP.TRAIN_REPLAY = False
        net.train_step_async(sess, dict(shape=(1, 4, 6, 4), G=3), op)
        assert net.replay_stats == dict(eager=3, recorded=1, replayed=2)              # (the switch off: not counted, nothing replayed)
    finally:
        cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = old


def test_recording_is_dropped_when_the_derived_filter_set_grows(monkeypatch):
    """a recorded step re-derives exactly the filter images that existed when it was recorded.  When the set grows
    afterwards (a TEST-mode network's first run on the shared session adds operand planes; another shape adds a plan key), the recording
    no longer covers it: the next step of that shape runs EAGERLY (weights_changed + refresh see the whole set), the one after it is
    recorded again, then replays resume.  Also: a recording starts with the stream's tier waits forgotten (every tier's first use is
    recorded), and two solver handles never share a recording."""
    from model.config import cfg
    log = []
    net, main, side, made = _stub_net(monkeypatch, log)
    sess, op = _Sess(), _TrainOp()
    old = (cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS)
    cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = True, 0
    blobs = dict(shape=(1, 4, 6, 4), G=3)
    try:
        for _ in range(3):
            net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=1, recorded=1, replayed=1) and sess.prepared.forgot == ["main"]
        sess.derived_gen += 1                                   # e.g. Session.h2_planes added an entry for a TEST-mode network
        net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=1, recorded=2, replayed=1)      # steady state still holds -> recorded afresh right away
        net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=1, recorded=2, replayed=2)
        sess.prepared.gen += 1                                  # a plan key added by another shape's eager step
        sess.prepared.version += 1                              # ... whose solver step has not refreshed yet: not steady
        net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=2, recorded=2, replayed=2)
        sess.prepared.ready_version = sess.prepared.version
        net.train_step_async(sess, blobs, op)
        net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=2, recorded=3, replayed=3)
        # a step that derives an image WHILE it is recorded is not the steady state: kept eager
        body = type(net)._train_step_body

        def growing(self, sess_, train_op, out):
            sess_.derived_gen += 1
            return body(self, sess_, train_op, out)
        monkeypatch.setattr(type(net), "_train_step_body", growing)
        sess.derived_gen += 1                                   # the recording is stale -> the step runs (and would be recorded) eagerly, growing the set
        net.train_step_async(sess, blobs, op)
        assert net.replay_stats == dict(eager=3, recorded=3, replayed=3)
        monkeypatch.setattr(type(net), "_train_step_body", body)
        op2 = _TrainOp()                                        # a second solver handle with the same signature: its own entry
        net.train_step_async(sess, blobs, op2)
        assert net.replay_stats["replayed"] == 3 and len([k for k in sess.graphs if k[0] == "train_replay"]) == 2
    finally:
        cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = old


def test_evicting_the_entry_that_runs_the_stream_search_ends_the_search(monkeypatch):
    """the StreamPicker belongs to ONE shape's entry; when that entry is evicted (REPLAY_CAP) or its recording dropped while
    the search is running, sess.picking is cleared so that another recording can start a search."""
    from model.config import cfg
    log = []
    net, main, side, made = _stub_net(monkeypatch, log)
    sess, op = _Sess(), _TrainOp()
    old = (cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS, type(net).REPLAY_CAP)
    cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS, type(net).REPLAY_CAP = True, 2, 2
    try:
        a, b, c = (dict(shape=(1, 4, w, 4), G=3) for w in (6, 8, 10))
        for _ in range(3):
            net.train_step_async(sess, a, op)                   # eager, recorded, first replay: the search starts on shape a
        assert sess.picking and getattr(sess, "picked_streams", None) is None
        net.train_step_async(sess, b, op)
        net.train_step_async(sess, b, op)
        assert sess.picking                                     # (shape a's entry is still there)
        net.train_step_async(sess, c, op)                       # third shape: the least recently used entry -- a's -- goes
        assert not sess.picking and len([k for k in sess.graphs if k[0] == "train_replay"]) == 2
        net.train_step_async(sess, b, op)                       # b's recording may start its own search now
        assert sess.picking
        sess.derived_gen += 1                                   # ... and dropping b's recording ends that one too
        net.train_step_async(sess, b, op)
        assert not sess.picking
    finally:
        cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS, type(net).REPLAY_CAP = old


def test_stream_picker_times_real_steps_and_keeps_the_fastest_binding(monkeypatch):
    """cfg.HIP.TRAIN_PICK_STREAMS = 2: baseline window + one window per (helper slot, pool stream), every window between two
    synchronisations of the main stream; the binding with the shortest window wins and later steps run on it."""
    import time
    from model.config import cfg
    log = []
    net, main, side, made = _stub_net(monkeypatch, log)
    sess, op = _Sess(), _TrainOp()
    cost = {0x2000: 0.004, 0x9000: 0.001, 0x9010: 0.006}                              # seconds a step "takes" with the helper slot on that stream
    real = FakeFn.__call__

    def slow(self, *args):
        if args[0].value == 0xC0:
            time.sleep(cost[args[-1].value])
        return real(self, *args)
    monkeypatch.setattr(FakeFn, "__call__", slow)
    old = (cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS)
    cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = True, 2
    try:
        for i in range(2 + 3 * 3 + 2):
            net.train_step_async(sess, dict(shape=(1, 4, 6, 4), G=3), op)
        assert [w for w, _ in sess.pick_log] == ["inherited", "slot 1 -> pool[0]", "slot 1 -> pool[1]"]
        assert [int(s.cuda_stream) for s in sess.picked_streams] == [0x9000] and not sess.picking
        del log[:]
        net.train_step_async(sess, dict(shape=(1, 4, 6, 4), G=3), op)
        assert [e for e in log if e[0] == "frcnn_conv"][-1] == ("frcnn_conv", 0xC0, 9, 0x9000)
        assert net.replay_stats["replayed"] == 3 * 3 + 2 + 1
    finally:
        cfg.HIP.TRAIN_REPLAY, cfg.HIP.TRAIN_PICK_STREAMS = old
