"""One training step as a flat launch list (the stand-in for the reference's ONE `sess.run` per step, lib/nets/network.py:488-498,
lib/model/train_val.py:260-300).

The step the Python host code enqueues is ~1 300 C-ABI launches + ~350 event records / waits over 4-5 HIP streams, and every argument
of every launch is a function of the image shape only: static session buffers, static arena buffers (below), constants.  What costs the
host 11 us per launch is not the launch but the Python around it -- dictionary lookups of buffers, shape arithmetic, ctypes argument
conversion, the sweep's bookkeeping.  A `Recording` is that step with the Python taken out: while ONE eager step runs, every `call()` of
the binding (frcnn_hip.call) and every stream-level operation of the step (ops.ev_record / st_wait_event / st_wait_stream / t_copy /
t_zero / host_op) is appended to a list; later steps of the same shape REPLAY the list -- the same launches with the same arguments on
the same streams in the same order, hence the same bits as the eager step by construction -- in one tight loop.

Streams are recorded as SLOTS (0 = the stream the step was started on, 1.. = helper streams in order of first use), so a recording can be
replayed with another assignment of physical streams to slots (`Recording.bind`): which hardware queue a helper stream lands on is worth
milliseconds per step, and `StreamPicker` chooses the assignment by timing real steps at start-up.

The few arguments that do change from step to step (the sampling seeds of the target layers, dropout seeds) are marked when they are
recorded (`patch_last`) and rewritten at replay from the step's seed counter.

"""
import ctypes

import torch


class Arena(object):
    """Static result tensors for the wrappers in frcnn_hip.ops that allocate their outputs (`torch.empty` inside an op): while an arena
    is active (ops.arena), the n-th allocation of a step returns the session buffer ("arena", tag, n, shape, dtype) -- the same address
    every step, so a recorded launch that wrote it can be replayed, and a later eager step of the same shape reuses it."""

    def __init__(self, sess, tag):
        self.sess, self.tag, self.n = sess, tag, 0

    def reset(self):
        self.n = 0

    def take(self, shape, dtype, device):
        key = ("arena", self.tag, self.n, tuple(int(s) for s in shape), dtype)
        self.n += 1
        store = self.sess._store() if hasattr(self.sess, "_store") else self.sess.buffers      # (per image shape inside Session.shape_scope)
        t = store.get(key)
        if t is None:
            t = store[key] = torch.empty(key[3], dtype=dtype, device=self.sess.device if device is None else device)
        return t


class Recording(object):
    CALL, OP = 0, 1

    def __init__(self, main_stream):
        self.main = main_stream
        self.slot_of = {int(main_stream.cuda_stream): 0}        # stream handle -> slot
        self.streams = [main_stream]                            # slot -> torch stream it was recorded on
        self.cmds = []                                          # (CALL, fn, [ctypes args], stream arg index or -1, slot) | (OP, closure(rec))
        self.patches = []                                       # (command index, argument index, variable, value at record time, multiplier, ctype)
        self.vars = dict(seed=0, gt=0)                          # the recorded step's own variables
        self.keep = []                                          # whatever the recorded arguments point to and nobody else holds
        self.bound = None                                       # slot -> torch stream of the compiled program
        self.prog = None
        self.n_calls = 0

    # ---- recording ---------------------------------------------------------------------------------------------------------------------
    def slot(self, stream):
        """slot of a torch stream (helper streams register themselves at first use)"""
        h = int(stream.cuda_stream)
        s = self.slot_of.get(h)
        if s is None:
            s = self.slot_of[h] = len(self.streams)
            self.streams.append(stream)
        return s

    def _slot_of_handle(self, h):
        s = self.slot_of.get(h)
        if s is None:
            s = self.slot_of[h] = len(self.streams)
            self.streams.append(torch.cuda.ExternalStream(h))
        return s

    def add_call(self, fn, name, args):
        if name in HOST_ONLY:                                   # host arithmetic on host arrays: nothing a later step has to repeat
            return
        conv = []
        types = fn.argtypes or []
        for i, a in enumerate(args):
            if isinstance(a, ctypes._SimpleCData) or isinstance(a, ctypes.Array) or i >= len(types):
                conv.append(a)
            else:
                conv.append(types[i](a))
        si, slot = -1, -1
        if conv and isinstance(conv[-1], ctypes.c_void_p) and name not in NO_STREAM_ARG:
            h = int(conv[-1].value or 0)                        # every launch entry takes its stream last (include/frcnn_hip.h)
            if h or 0 in self.slot_of:                          # (0: the legacy default stream, when the step itself runs on it)
                si, slot = len(conv) - 1, self._slot_of_handle(h)
        self.cmds.append((self.CALL, fn, conv, si, slot, name))
        self.n_calls += 1

    def patch_last(self, arg_index, mult=1, var="seed"):
        """The last recorded call's argument `arg_index` changes from step to step.  var = "seed": a sampling seed, replay adds
        mult * (the step's seed - the recorded step's seed); var = "gt": the number of ground-truth boxes of the step's image."""
        i = len(self.cmds) - 1
        a = self.cmds[i][2][arg_index]
        self.patches.append((i, arg_index, var, int(a.value), int(mult), type(a)))

    def add_op(self, closure):
        """closure(rec): a stream-level operation of the step (event record / wait, tensor copy, collective); streams by rec.stream(slot)"""
        def op(rec, closure=closure):                           # (whatever the operation returns -- a tensor, a work handle -- is not a status code)
            closure(rec)
        self.cmds.append((self.OP, op))

    # ---- replay ------------------------------------------------------------------------------------------------------------------------
    def stream(self, slot):
        return self.bound[slot]

    def bind(self, streams):
        """compile the list for an assignment slot -> torch stream (slot 0 = the stream the step runs on)"""
        streams = list(streams)
        assert len(streams) == len(self.streams)
        self.bound = streams
        handles = [ctypes.c_void_p(int(s.cuda_stream)) for s in streams]
        prog = []
        for c in self.cmds:
            if c[0] == self.CALL:
                _, fn, conv, si, slot, _name = c
                if si >= 0:
                    conv = list(conv)
                    conv[si] = handles[slot]
                prog.append((fn, tuple(conv)))
            else:
                prog.append((c[1], (self,)))
        self.prog = prog
        return self

    def bound_to(self, main_stream):
        return self.bound is not None and int(self.bound[0].cuda_stream) == int(main_stream.cuda_stream)

    def default_binding(self, main_stream):
        return [main_stream] + list(self.streams[1:])

    def _patch(self, now):
        for i, ai, var, val, mult, typ in self.patches:
            fn, args = self.prog[i]
            a = list(args)
            a[ai] = typ(val + mult * (int(now["seed"]) - int(self.vars["seed"]))) if var == "seed" else typ(int(now[var]))
            self.prog[i] = (fn, tuple(a))

    def replay(self, now):
        """Enqueue the step.  now: this step's variables {"seed": Network._sample_seed, "gt": rows of gt_boxes}; torch's current stream
        must be the stream bound to slot 0."""
        if self.patches:
            self._patch(now)
        for fn, args in self.prog:
            rc = fn(*args)
            if rc:
                from . import check
                check(rc, getattr(fn, "__name__", "recorded call"))


class StreamPicker(object):
    """Which PHYSICAL stream each helper slot of a recorded step runs on, chosen by timing real steps.

    ROCm multiplexes HIP streams onto a few hardware queues (4 by default), and whether a helper stream of the step -- filter gradients,
    solver, filter preparation -- shares a queue with the data-gradient chain it is meant to run beside is decided by the order in which
    streams were created in the process: an RCCL process group that merely EXISTS moved the step from 19.4 to 23.4 ms in round 4
    (profiles/r04_ao_dp_pieces.txt), GPU_MAX_HW_QUEUES=8 gave that back and cost the single-GPU step 24 ms.  A recording names streams by
    slot, so the assignment can be searched instead of inherited: one pass of coordinate descent -- for every helper slot in turn, every
    stream of a small pool -- each candidate timed over a window of `window` real training steps between two device synchronisations, a
    candidate kept when it beats the best so far by more than 1 %.  Any binding is CORRECT (the list's events order the streams whatever
    they are; two slots on one stream merely serialise), so the search cannot change a bit of the result -- the steps it times are
    ordinary training steps."""

    def __init__(self, rec, main, pool, window=3, gain=0.01):
        self.rec, self.pool, self.window, self.gain = rec, list(pool), int(window), float(gain)
        self.best = rec.default_binding(main)
        self.best_s = None
        self.queue = [None]                                  # None = the inherited binding (the baseline), then (slot, pool index)
        self.queue += [(s, i) for s in range(1, len(self.best)) for i in range(len(self.pool))]
        self.left, self.t0, self.cur = 0, 0.0, None
        self.done = len(self.best) < 2
        self.log = []                                        # (what, seconds per step)

    def _binding(self, cand):
        b = list(self.best)
        if cand is not None:
            b[cand[0]] = self.pool[cand[1]]
        return b

    def before_step(self, main):
        import time
        if self.done or self.left:
            return
        while self.queue:
            self.cur = self.queue.pop(0)
            b = self._binding(self.cur)
            b[0] = main
            if self.cur is None or int(b[self.cur[0]].cuda_stream) != int(self.best[self.cur[0]].cuda_stream):
                break                                        # (a candidate equal to the slot's current stream was measured already)
        else:
            self._finish(main)
            return
        main.synchronize()
        self.rec.bind(b)
        self.left, self.t0 = self.window, time.perf_counter()

    def after_step(self, main):
        import time
        if self.done or not self.left:
            return
        self.left -= 1
        if self.left:
            return
        main.synchronize()
        s = (time.perf_counter() - self.t0) / self.window
        self.log.append(("inherited" if self.cur is None else "slot %d -> pool[%d]" % self.cur, s))
        if self.best_s is None or s < self.best_s * (1.0 - self.gain):
            self.best_s = s
            if self.cur is not None:
                self.best[self.cur[0]] = self.pool[self.cur[1]]
        if not self.queue:
            self._finish(main)

    def _finish(self, main):
        b = list(self.best)
        b[0] = main
        self.rec.bind(b)
        self.done = True


# Entries of include/frcnn_hip.h WITHOUT a trailing `void* stream` (tests/test_replay_cpu.py checks the two sets against the header):
# launch-context setters (thread-local state the following launches read: recorded, replayed in order) ...
NO_STREAM_ARG = frozenset(["frcnn_set_tuning", "frcnn_detect_set_tuning", "frcnn_conv2d_wgrad_set_plan", "frcnn_conv2d_wgrad_h2_set_plan"])
# ... and host functions on host memory (their pointer arguments do not outlive the call: never recorded)
HOST_ONLY = frozenset(["frcnn_generate_anchors", "frcnn_winograd_filter_transform", "frcnn_winograd7_filter_transform", "frcnn_pack_filter_hwio",
                       "frcnn_prep_image_shape", "frcnn_crc32c", "frcnn_snappy_uncompress", "_nms", "frcnn_graph_begin", "frcnn_graph_end",
                       "frcnn_graph_launch", "frcnn_graph_destroy", "frcnn_abi_version", "frcnn_build_info", "frcnn_sgd_desc_bytes",
                       "frcnn_conv2d_wgrad_supported"])
