"""Training step on the device chain (lib/model/train_val.py:116-153,
lib/nets/network.py:488-516): reverse sweep over the tape the Network records in TRAIN mode, parameter
gradients, momentum SGD with slim-style L2 regularisation, optional data-parallel gradient all-reduce.

Convolution gradients run on the SAME f32-MFMA implicit-GEMM kernel as the forward pass, on re-laid-out
operands (csrc/backward_kernels.hip).  BN is frozen in the reference (resnet_v1.py:22-44), so the
forward pass uses BN-folded filters; the master copy of every filter is kept un-folded on the device in
the packed layout [Cout][KH][KW][Cin], its gradient is dW_folded * scale[n] (chain rule through the
fold) and the folded copy is refreshed inside the SGD kernel.
"""
import numpy as np
import torch

from . import ACT_NONE, ACT_RELU, ACT_RELU6, ops


class Param(object):
    """One trainable filter (+ optional bias) on the device."""

    def __init__(self, scope, wf, bias, scale_np, bias_trainable, flat_w, flat_b, master_hwio=None):
        dev = wf.device
        self.scope = scope
        self.wf = wf                                            # folded filter used by the forward kernels
        self.K = wf[0].numel()
        self.scale = None if scale_np is None else torch.from_numpy(np.ascontiguousarray(scale_np, dtype=np.float32)).to(dev)
        if self.scale is None:
            self.w = wf                                         # no fold: master == forward filter
        elif master_hwio is not None and master_hwio.size == wf.numel():
            # the un-folded variable itself (exactly what a snapshot stores), and the folded copy recomputed from it with the
            # SGD kernel's own f32 product, so a resumed run starts from the very state the interrupted one had
            m = np.ascontiguousarray(master_hwio.reshape((wf.shape[1], wf.shape[2], wf.shape[3], wf.shape[0])).transpose(3, 0, 1, 2))
            self.w = torch.from_numpy(m.astype(np.float32)).to(dev)
            wf.copy_(self.w * self.scale.view(-1, 1, 1, 1))
        else:
            self.w = (wf / self.scale.view(-1, 1, 1, 1)).contiguous()
        self.acc_w = torch.zeros_like(self.w)
        self.grad_w = flat_w.view(wf.shape)                     # view into the flat gradient buffer (one all-reduce)
        self.bias = bias if bias_trainable else None
        self.acc_b = torch.zeros_like(bias) if bias_trainable else None
        self.grad_b = flat_b


class DwParam(object):
    """One trainable depthwise 3x3 filter (MobileNet): master copy [3,3,C] as in the variable `depthwise_weights` [3,3,C,1], the
    forward filter is master * scale[c] (frozen-BN fold, refreshed after every solver step); no bias (the BN shift is frozen)."""
    dw = True

    def __init__(self, scope, wf, scale_np, flat_w, master_np):
        dev = wf.device
        self.scope = scope
        self.wf = wf
        self.scale = torch.from_numpy(np.ascontiguousarray(scale_np, dtype=np.float32)).to(dev)
        self.w = torch.from_numpy(np.ascontiguousarray(master_np, dtype=np.float32)).to(dev)
        ops.dwconv3x3_refold(self.w, self.scale, self.wf)
        self.K = self.w.numel()
        self.acc_w = torch.zeros_like(self.w)
        self.grad_w = flat_w.view(self.w.shape)
        self.bias = self.acc_b = self.grad_b = None


class TrainState(object):
    """Owns the parameters, their momentum buffers and ONE flat gradient buffer (so data-parallel
    training is a single RCCL all-reduce, or a few bucketed ones, over contiguous memory)."""

    def __init__(self, sess, net, momentum=0.9, weight_decay=1e-4, double_bias=False, bias_decay=False):
        self.sess, self.net = sess, net
        self.momentum, self.weight_decay = momentum, weight_decay
        self.double_bias, self.bias_decay = double_bias, bias_decay
        self.params = {}
        self.flat = None
        self.reg_scopes = []
        self._wgrad_events = None
        self.fuse_chain = True                                  # see _sweep (False: separate relu_bwd / copy / h2_split passes; tests)
        self.pipe_dgrads = True                                 # see _sweep (False: frcnn_conv2d_dgrad_strided for every strided / odd-width layer; tests)
        self.prep_stream = True                                 # sess.prepared: weight-only launches re-run by the solver on a side stream

    def build(self):
        """Call after one TRAIN forward (which packs every filter and fills net._tape)."""
        sess, net = self.sess, self.net
        scopes = []                                             # (kind, scope) in forward order: the flat buffer follows the tape
        for rec in net._tape:
            if rec["kind"] in ("conv", "dwconv") and net.trainable_scope(rec["scope"]):
                if (rec["kind"], rec["scope"]) in scopes:
                    # one tape record per trainable filter: the reverse sweep OVERWRITES grad_w per record, and the overlapped
                    # all-reduce (parallel.BucketedAllReduce) ships a gradient range as soon as its record has been visited
                    raise NotImplementedError("trainable scope %s is used by two layers of the TRAIN graph (shared filters)" % rec["scope"])
                scopes.append((rec["kind"], rec["scope"]))
        total = 0
        sizes = []
        for kind, sc in scopes:
            if kind == "dwconv":
                sizes.append((sess.packed[("dw", sc)][0].numel(), 0))
            else:
                info = sess.conv_info[sc]
                sizes.append((info["w"].numel(), info["b"].numel() if (info["b"] is not None and not info["bn"]) else 0))
            total += sum(sizes[-1])
        self.flat = torch.zeros((total,), dtype=torch.float32, device=sess.device)
        off = 0
        for (kind, sc), (nw, nb) in zip(scopes, sizes):
            fw = self.flat[off:off + nw]
            off += nw
            fb = self.flat[off:off + nb] if nb else None
            off += nb
            if kind == "dwconv":
                scale, _ = sess.fold_bn(sc, net.dw_bn_eps)
                self.params[sc] = DwParam(sc, sess.packed[("dw", sc)][0], scale, fw, sess.variables[sc + "/depthwise_weights"][:, :, :, 0])
            else:
                info = sess.conv_info[sc]
                self.params[sc] = Param(sc, info["w"], info["b"], info["scale"], nb > 0, fw, fb, sess.variables.get(sc + "/weights"))
        self.reg_scopes = [sc for sc in sess.conv_info]            # slim regularises every conv/fc weight, frozen or not
        return self

    # ---- gradients -----------------------------------------------------------------------------------
    def data_parallel(self):
        """True when the step runs under the data-parallel rules: more than one replica, or `force_dp` (bench.py --dp-constrained: one
        replica with everything a multi-GPU step does -- at most one filter-gradient side stream, the bucketed all-reduce issued from inside
        the sweep over a one-rank RCCL group, no captured sweep -- so that the step an 8-GPU run executes per GPU can be timed at N = 1)."""
        return int(getattr(self, "world_size", 1)) > 1 or bool(getattr(self, "force_dp", False))

    def count_flops(self, pipe, flops):
        """Ledger of the reverse sweep's matrix work per matrix pipe ("h2": frcnn_gemm_h2 / frcnn_conv2d_wgrad_h2, "f32": the f32-MFMA
        kernels); bench.py prices a training step per pipe with it (host arithmetic on launch shapes, nothing on the device)."""
        led = getattr(self, "flop_ledger", None)
        if led is not None:
            led[pipe] = led.get(pipe, 0) + int(flops)

    def backward(self, seeds, fuse_solver=False):
        """seeds: list of (tensor, grad) for network outputs.  Fills every Param.grad_*.

        fuse_solver = True (what Network.train_step_async passes, and nobody else should): the sweep ALSO runs the solver for the tail of
        the parameter table on its own stream as the filter gradients finish (frcnn_sgd_momentum_range with TrainState.lr) -- weights and
        momentum of those parameters are updated when this returns, and the caller MUST complete the step with apply(self.lr, ...), which
        updates the rest.  With the default, backward() only computes gradients: calling it twice, inspecting gradients, accumulating
        them or skipping a step leaves the parameters alone."""
        if getattr(self, "_sgd_done_from", None) is not None:
            raise RuntimeError("TrainState.backward: the previous sweep updated part of the parameters inside the sweep (fuse_solver=True) "
                               "and apply() never completed that step")
        main = torch.cuda.current_stream()
        with ops.pinned_stream(main):                 # ~1000 launches: one stream lookup instead of one per launch
            return self._sweep(seeds, main, bool(fuse_solver))

    def replay_signature(self):
        """Everything about this solver handle that decides WHICH launches a step makes (a recorded step is valid for one signature)."""
        ar = getattr(self, "all_reduce", None)
        return (float(self.momentum), float(self.weight_decay), bool(self.double_bias), bool(self.bias_decay), bool(self.fuse_chain),
                bool(self.pipe_dgrads), bool(self.prep_stream), bool(getattr(self, "solver_in_sweep", True)), int(getattr(self, "world_size", 1)),
                bool(getattr(self, "force_dp", False)), None if ar is None else id(ar))

    def _sweep(self, seeds, main, fuse_solver=False):
        sess, net = self.sess, self.net
        grads = {}
        for t, g in seeds:
            grads[t.data_ptr()] = g
        needs = net._requires_grad
        # Filter gradients on side streams (cfg.HIP.WGRAD_STREAM = how many): a wgrad only feeds the solver, while the data-gradient
        # chain is what the next record waits for -- at one image per step neither fills 256 CUs, so they run side by side.  Each side
        # stream executes its wgrads in tape order with its OWN transposed-operand scratch and split-K workspace, sees dY through an
        # event recorded after the activation gradient, and is joined before backward() returns.  Data parallel: one side stream, so
        # that "everything behind this offset of the flat gradient is final" holds on the stream the collective is issued from.
        nside = int(getattr(self, "wgrad_stream", 0))
        ar = getattr(self, "all_reduce", None)
        dp = self.data_parallel() and ar is not None and hasattr(ar, "ready")        # the bucketed, overlapped exchange (parallel.BucketedAllReduce)
        if self.data_parallel() and not (dp and getattr(ar, "multi_stream", False)):
            nside = min(nside, 1)                                # a plain all-reduce callable orders itself after ONE stream
        sides = self._wgrad_side_streams(nside)
        turn = [0]
        if self._wgrad_events is None:
            self._wgrad_events = [torch.cuda.Event() for _ in range(16)]

        pins = [ops.pinned_stream(st) for st in sides]
        events = self._wgrad_events

        # The solver inside the sweep (single-GPU runs with side streams; from the second step on, when apply() has built its descriptor
        # table): the reverse sweep finishes the parameters from the END of the flat gradient backwards, so every SOLVER_CHUNK filter
        # gradients the solver stream updates that range -- frcnn_sgd_momentum_range -- behind (a) the side streams' filter gradients
        # enqueued so far and (b) the main stream's data gradients enqueued so far (the last readers of those filters in this step).
        # ~0.5 ms of memory-bound update leaves the tail of the step; apply() updates what is left (the first layers).  The L2
        # regulariser value reads the filters BEFORE any update: it opens the solver stream's step.
        self._sgd_done_from = None
        solver = None
        if (fuse_solver and sides and (not self.data_parallel() or (dp and getattr(ar, "multi_stream", False)))
                and getattr(self, "_sgd_table", None) is not None and getattr(self, "lr", None) is not None
                and getattr(self, "solver_in_sweep", True) and not any(getattr(p, "dw", False) for p in self.params.values())):
            solver = self._solver_stream_obj()
            ops.st_wait_stream(solver, main)                     # (the previous step's apply() -- nothing else of this step matters to it)
            with ops.pinned_stream(solver):
                self.regularization_loss(self._reg_total())
            self._reg_in_sweep = True
            self._sgd_done_from = self._sgd_count
        pending = [0]                                            # filter gradients enqueued since the last solver launch

        gs = 1.0 / float(getattr(self, "world_size", 1)) if self.data_parallel() else 1.0     # the mean over replicas, folded into the update

        def solver_step(first):
            """update table[first, done_from) on the solver stream"""
            while len(self._solver_events) < 1 + len(sides):     # (cfg.HIP.WGRAD_STREAM is not bounded)
                self._solver_events.append(torch.cuda.Event())
            for i, st in enumerate([main] + list(sides)):
                ev = self._solver_events[i]
                ops.ev_record(ev, st)
                ops.st_wait_event(solver, ev)
            if dp:
                ar.wait_sent(solver)                             # ... and behind the all-reduces that sum these gradients over the replicas
            with ops.pinned_stream(solver):
                ops.sgd_momentum_range(self._sgd_table, first, self._sgd_done_from - first, self.lr, self.momentum, gs)
            self._sgd_done_from = first
            pending[0] = 0

        def dp_solver_step():
            """data parallel: every parameter whose gradient lies inside the range already handed to the all-reduce ([sent_from, end) of
            the flat buffer; the descriptor table is in the buffer's order) can be updated behind that collective"""
            if solver is None or ar.sent_from is None:
                return
            import bisect
            first = bisect.bisect_left(self._sgd_offsets, int(ar.sent_from))
            if first < self._sgd_done_from:
                solver_step(first)

        def on_side(fn):
            if not sides:
                return fn("", None)
            i = turn[0] % len(sides)
            ev = events[turn[0] % len(events)]                   # a small ring: a wait captures the record that precedes it
            turn[0] += 1
            side = sides[i]
            ops.ev_record(ev, main)
            ops.st_wait_event(side, ev)
            scope, ops.ws_scope = ops.ws_scope, ops.ws_scope + "/wgrad%d" % i
            try:
                with pins[i]:                                    # this module's launches go to `side` without a torch stream switch; a
                    fn("/s%d" % i if i else "", side)            # bucket that becomes ready is sent with `side` current (parallel.py)
            finally:
                ops.ws_scope = scope

        # Weight-only launches of the data-gradient chain (transposed / flipped filters, their h2 split, the Winograd transform of
        # the gradient filter) go through sess.prepared (runtime.PreparedFilters): after the first step the solver re-runs them on a
        # side stream right after its update, beside the next forward pass, and this sweep finds them done.
        prep = sess.prepared
        prep.enabled = bool(getattr(self, "prep_stream", True)) and not getattr(self, "graph", False)      # (a captured sweep prepares inline)
        prepared = prep.get

        # Elementwise passes of the chain rule folded into the launches on either side of them (cfg-free; `fuse_chain = False` keeps the
        # separate passes, for the bit-equality test):
        #  * ReLU gradient: the input x of a trunk convolution is a ReLU output, so the data gradient a record produces is masked by
        #    (x > 0) in the producing launch's epilogue (ops.*(mask=x)); `masked` = tensors whose gradient buffer holds ONLY masked
        #    contributions -- their producer record skips its relu_bwd pass (the select is idempotent and distributes over the sum, so a
        #    buffer that also received an unmasked contribution is simply masked again by the pass);
        #  * identity shortcut: the residual's gradient IS the masked dY of the unit's last convolution -- `borrowed` maps it to that
        #    buffer instead of copying it; the convolution that later adds its own data gradient reads it as the launch's residual and
        #    writes a fresh buffer (the borrowed one is still being read by the filter gradient on a side stream);
        #  * operand planes of dY for a following frcnn_gemm_h2 data gradient come out of the Winograd output transform (`emitted`).
        fuse = bool(getattr(self, "fuse_chain", True))
        pipe = bool(getattr(self, "pipe_dgrads", True))         # strided 3x3 / odd-width 1x1 data gradients on the matrix pipe (False: gather kernel)
        producer = {}
        if fuse:
            for r in net._tape:
                if r["kind"] not in ("mean", "crop", "maxpool", "dropout", "dwconv"):
                    producer[r["y"].data_ptr()] = r
        masked, borrowed, emitted = set(), set(), {}

        def relu_mask(t):
            r = producer.get(t.data_ptr())
            return r["y"] if (r is not None and r["act"] == ACT_RELU) else None

        def accumulate_into(target, shape, name):
            key = target.data_ptr()
            if key in grads:
                emitted.pop(key, None)                       # (planes emitted with the first contribution would be stale)
                if key in borrowed:                          # an in-place accumulation is about to write it: take a private copy
                    g = sess.buf("grad/" + name + "/own", shape)
                    ops.t_copy(g, grads[key].view(shape))
                    grads[key] = g
                    borrowed.discard(key)
                masked.discard(key)                          # (callers that mask the whole sum add it again)
                return grads[key], True
            g = sess.buf("grad/" + name, shape)
            grads[key] = g
            masked.discard(key)
            return g, False

        def h2_dgrad(r):
            """Does the record's data gradient run as frcnn_gemm_h2 (dX = dY W, K = Cout)?  (cfg.HIP.H2_TRAIN)"""
            if r["k"] != 1 or r["stride"] != 1 or tuple(r["pad"]) != (0, 0, 0, 0) or getattr(self, "h2_train", None) is None:
                return False
            yy, wf = r["y"], sess.conv_info[r["scope"]]["w"]
            Mr, Co, Ci = yy.numel() // yy.shape[-1], yy.shape[-1], wf.shape[3]
            return (Co % 128 == 0 and Ci % 128 == 0 and ((Mr + 127) // 128) * (Ci // 128) >= self.h2_train
                    and 4 * Mr * Co < (1 << 32) and Mr * Ci < (1 << 29))

        for rec in reversed(net._tape):
            kind = rec["kind"]
            if kind == "mean":
                gy = grads.get(rec["y"].data_ptr())
                if gy is None:
                    continue
                x = rec["x"]
                gx, had = accumulate_into(x, x.shape, rec["name"] + "/in")
                assert not had
                ops.spatial_mean_bwd(gy.view(rec["y"].shape), x.shape[1] * x.shape[2], gx)
                continue
            if kind == "crop":
                gy = grads.get(rec["y"].data_ptr())
                if gy is None or rec["feat"].data_ptr() not in needs:
                    continue
                feat = rec["feat"]
                gx, had = accumulate_into(feat, feat.shape, "feat")
                if not had:
                    ops.t_zero(gx)
                ops.crop_and_resize_bwd(gy.view(rec["y"].shape), rec["rois"], rec["stride"], gx)
                continue
            if kind == "maxpool":
                gy = grads.get(rec["y"].data_ptr())
                x = rec["x"]
                if gy is None or x.data_ptr() not in needs:
                    continue
                gx, had = accumulate_into(x, x.shape, rec["name"] + "/in")
                assert not had
                ops.maxpool_bwd(x, rec["y"], gy.view(rec["y"].shape), rec["k"], rec["stride"], gx)
                continue
            if kind == "dropout":
                gy = grads.get(rec["y"].data_ptr())
                x = rec["x"]
                if gy is None or x.data_ptr() not in needs:
                    continue
                gx, had = accumulate_into(x, x.shape, rec["name"] + "/in")
                assert not had
                ops.dropout(gy.view(x.shape), rec["seed"], rec["keep"], out=gx, step_mult=256)      # same mask, same 1 / keep_prob
                continue
            if kind == "dwconv":
                y, x, sc = rec["y"], rec["x"], rec["scope"]
                gy = grads.get(y.data_ptr())
                if gy is None:
                    continue
                gy = gy.view(y.shape)
                (ops.relu6_bwd if rec["act"] == ACT_RELU6 else ops.relu_bwd)(gy, y)
                p = self.params.get(sc)
                if p is not None:
                    def dw_wgrad(sfx, side, gy=gy, x=x, rec=rec, p=p):
                        ops.dwconv3x3_wgrad(gy, x, rec["stride"], rec["pad"], p.scale, p.grad_w)
                        if dp:
                            ar.ready(self.flat, p.grad_w.data_ptr(), side, sides)
                    on_side(dw_wgrad)
                if x.data_ptr() in needs:
                    gx, had = accumulate_into(x, x.shape, sc + "/in")
                    ops.dwconv3x3_dgrad(gy, sess.packed[("dw", sc)][0], rec["stride"], rec["pad"], gx, accumulate=had)
                continue
            # ---- convolution record -------------------------------------------------------------------
            y, x, sc = rec["y"], rec["x"], rec["scope"]
            gy = grads.get(y.data_ptr())
            if gy is None:
                continue
            gy = gy.view(y.shape)
            if rec["act"] != ACT_NONE and y.data_ptr() in borrowed:      # the activation gradient is applied in place
                gy = sess.buf("grad/" + sc + "/own", y.shape)
                ops.t_copy(gy, grads[y.data_ptr()].view(y.shape))
                grads[y.data_ptr()] = gy
                borrowed.discard(y.data_ptr())
            if rec["act"] == ACT_RELU:
                if y.data_ptr() not in masked:
                    ops.relu_bwd(gy, y)
            elif rec["act"] == ACT_RELU6:
                ops.relu6_bwd(gy, y)
            res = rec["residual"]
            if res is not None and res.data_ptr() in needs:
                if fuse and rec["res_stride"] == 1 and res.data_ptr() not in grads:
                    grads[res.data_ptr()] = gy                   # identity shortcut: no copy (see `borrowed` above)
                    borrowed.add(res.data_ptr())
                    masked.discard(res.data_ptr())
                else:
                    gr, had = accumulate_into(res, res.shape, sc + "/res")
                    if rec["res_stride"] == 1 and not had:
                        ops.t_copy(gr, gy)
                    else:
                        if not had:
                            ops.t_zero(gr)
                        ops.add_strided(gy, gr, rec["res_stride"], True)
            k, stride, pad = rec["k"], rec["stride"], rec["pad"]
            N, OH, OW, Cout = y.shape
            M = N * OH * OW
            p = self.params.get(sc)
            if p is not None:
                def wgrad(sfx, side, gy=gy, x=x, p=p, k=k, stride=stride, pad=pad, OH=OH, OW=OW, M=M, Cout=Cout):
                    if getattr(self, "wgrad_tn", True) and ops.conv2d_wgrad_supported(x.shape[-1], Cout) and p.K == k * k * x.shape[-1]:
                        # dW = dY^T X straight from the two tensors as they lie (csrc/wgrad_tn.hip): no transposed copies, no im2col
                        ops.conv2d_wgrad(gy, x, k, k, stride, pad, p.grad_w, h2=bool(getattr(self, "wgrad_h2", False)))
                        self.count_flops("h2" if getattr(self, "wgrad_h2", False) else "f32", 2 * M * Cout * p.K)
                        if p.bias is not None:
                            ops.colsum(gy.view(M, Cout), p.grad_b)
                        if dp:
                            ar.ready(self.flat, p.grad_w.data_ptr(), side, sides)
                        return
                    Mp = (M + 31) // 32 * 32
                    gyT = ops.transpose_pad(gy.view(M, Cout), Mp, out=sess.buf("bwd/gyT" + sfx, (Cout, Mp)))
                    if k == 1 and stride == 1:
                        xT = ops.transpose_pad(x.view(M, x.shape[-1]), Mp, out=sess.buf("bwd/xT" + sfx, (x.shape[-1], Mp)))
                    else:
                        xT = ops.im2col_t(x, k, k, stride, pad, OH, OW, Mp, out=sess.buf("bwd/xT" + sfx, (k * k * x.shape[-1], Mp)))
                    # dW_folded[n][(kh,kw,c)] = sum_m gyT[n][m] * xT[(kh,kw,c)][m]   -- the forward MFMA kernel
                    ops.conv2d(gyT.view(1, 1, Cout, Mp), xT.view(xT.shape[0], 1, 1, Mp), None, 1, 1, out=p.grad_w.view(1, 1, Cout, p.K))
                    self.count_flops("f32", 2 * M * Cout * p.K)
                    if p.bias is not None:
                        ops.colsum(gy.view(M, Cout), p.grad_b)
                    if dp:
                        # this parameter's gradient is enqueued: everything from its offset to the end of the flat buffer is final
                        # (the tape is walked backwards, the buffer is laid out in forward order; the collective orders itself after
                        # every side stream a filter gradient may have been enqueued on)
                        ar.ready(self.flat, p.grad_w.data_ptr(), side, sides)
                on_side(wgrad)
            if x.data_ptr() in needs:
                key = x.data_ptr()
                mk = relu_mask(x) if fuse else None             # x itself when it is a ReLU output: the gradient's mask
                lent = key in borrowed                          # an identity shortcut's gradient waits there, in somebody else's buffer
                wf = sess.conv_info[sc]["w"]
                wino = getattr(self, "winograd", None)          # (m, min channels) set by the Network from cfg.HIP, or None
                Cin = wf.shape[3]
                flipped = stride == 1 and Cout % 32 == 0
                up_h, up_w = (OH - 1) * stride + 1, (OW - 1) * stride + 1
                up_pad = (k - 1 - pad[0], x.shape[1] - up_h - (k - 1 - pad[0]) + k - 1, k - 1 - pad[2], x.shape[2] - up_w - (k - 1 - pad[2]) + k - 1)
                upsampled = pipe and stride > 1 and k > 1 and Cout % 32 == 0 and Cin % 32 == 0 and min(up_pad) >= 0
                padded = pipe and k == 1 and stride == 1 and tuple(pad) == (0, 0, 0, 0) and Cout % 32 != 0 and Cin % 4 == 0 and y.dim() == 4
                # (a tensor that already holds a gradient -- the RPN's 3x3 behind the crop's: Winograd into a scratch tensor + one add pass
                # instead of the direct kernel with its residual epilogue: 4608-deep float32 products, 490 us, r04_aj_train_streams.txt)
                use_wino = (wino is not None and (key not in grads or pipe) and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1)
                            and Cout % 32 == 0 and Cout >= wino[1] and Cin % 4 == 0)
                if lent and (use_wino or not (flipped or upsampled or padded)):
                    lent = False                                # (accumulate_into below takes the private copy)
                if lent:
                    # out-of-place: residual = the borrowed buffer (read only), result = this record's own buffer
                    gres, had = grads[key], True
                    emitted.pop(key, None)
                    gx = sess.buf("grad/" + sc + "/in", x.shape)
                    grads[key] = gx
                    borrowed.discard(key)
                    masked.discard(key)
                else:
                    gx, had = accumulate_into(x, x.shape, sc + "/in")
                    gres = gx
                if use_wino:
                    # dX = conv(dY, flipped / transposed filter) is itself a 3x3 stride-1 SAME convolution: Winograd, with the
                    # gradient filter transformed straight from the packed forward filter
                    m = 7 if (wino[0] == 4 and len(wino) > 2 and wino[2] and OH == 7 and OW == 7) else wino[0]
                    G = ops.winograd_points(m)
                    T = ops.winograd_tiles(N, OH, OW, m)
                    u = prepared(("wino_u", sc, m), lambda wf=wf, m=m, sc=sc, G=G, Cin=Cin, Cout=Cout: ops.winograd_filter_transform_device(
                        wf, m, True, out=sess.buf("bwd/wino_u/" + sc, (G, Cin, Cout))))
                    fused = mk is not None and m in (4, 7) and not had
                    gxp = None
                    if fused and Cin % 128 == 0 and h2_dgrad(producer[key]):
                        gxp = emitted[key] = sess.h2_buf("bwd/gyp/" + sc, x.numel() // Cin, Cin)      # dY planes of the producer's dgrad GEMM
                    dst = sess.buf("bwd/wino_sum", x.shape) if had else gx
                    ops.conv3x3_winograd(gy, u, None, ACT_NONE, out=dst, v_buf=sess.buf("bwd/wino_v", (G, T, Cout)),
                                         m_buf=sess.buf("bwd/wino_m", (G, T, Cin)), mask=mk if fused else None, out_planes=gxp)
                    if had:
                        ops.add_strided(dst, gx, 1, True)
                    if fused:
                        masked.add(key)
                    self.count_flops("f32", 2 * G * T * Cin * Cout)
                elif flipped:
                    wd = prepared(("wflip", sc), lambda wf=wf, sc=sc, k=k, Cout=Cout: ops.flip_transpose_filter(
                        wf, out=sess.buf("bwd/wflip/" + sc, (wf.shape[3], k, k, Cout))))
                    if h2_dgrad(rec):
                        # dX = dY W: a plain GEMM with K = Cout -- frcnn_gemm_h2 on the split of dY and of the transposed filter
                        # (cfg.HIP.H2_TRAIN; both change every step, so both are split here: 8 B per element of dY, a few MB of filter)
                        gp = emitted.pop(y.data_ptr(), None)
                        if gp is None:
                            gp = ops.h2_split(gy.view(M, Cout), out=sess.h2_buf("bwd/gy", M, Cout))
                        wq = prepared(("wflip_h2", sc), lambda wd=wd, sc=sc, Cin=Cin, Cout=Cout: ops.h2_pack_w(
                            wd.view(Cin, Cout), out=sess.buf_pair("bwd/wflip_h2/" + sc, Cin, Cout)))
                        gxp = None
                        if mk is not None and h2_dgrad(producer[key]):     # the producer's own data gradient reads this result as planes
                            gxp = emitted[key] = sess.h2_buf("bwd/gyp/" + sc, M, Cin)
                        ops.gemm_h2(gp, wq, 1, M, Cin, Cout, None, gres.view(M, Cin) if had else None, ACT_NONE, out=gx.view(M, Cin),
                                    mask=None if mk is None else mk.view(M, Cin), out_planes=gxp)
                        self.count_flops("h2", 2 * M * Cin * Cout)
                    else:
                        dpad = (k - 1 - pad[0], k - 1 - pad[1], k - 1 - pad[2], k - 1 - pad[3])
                        ops.conv2d(gy, wd, None, k, k, 1, dpad, ACT_NONE, gres if had else None, 1, out=gx, mask=mk)
                        self.count_flops("f32", 2 * M * Cin * Cout * k * k)
                    if mk is not None:
                        masked.add(key)                          # mask * (dX + what was there): the whole buffer is masked
                elif padded:
                    # 1x1 heads whose output width is no multiple of 32 (cls_score 81, bbox_pred 324, the RPN's 2A / 4A): dX = dY W on the
                    # matrix pipe with dY and the transposed filter zero-padded to the next multiple (the gather kernel: 60-120 us each)
                    Cp = (Cout + 31) // 32 * 32
                    wdp = prepared(("wflip_pad", sc), lambda wf=wf, sc=sc, Cin=Cin, Cout=Cout, Cp=Cp: ops.transpose_pad(
                        wf.view(Cout, Cin), Cp, out=sess.buf("bwd/wflip_pad/" + sc, (Cin, Cp))))
                    gyp = sess.buf("bwd/gypad/" + sc, (N, OH, OW, Cp), zero=True)      # columns >= Cout stay zero
                    ops.t_copy(gyp[..., :Cout], gy)
                    ops.conv2d(gyp, wdp.view(Cin, 1, 1, Cp), None, 1, 1, 1, (0, 0, 0, 0), ACT_NONE, gres if had else None, 1, out=gx, mask=mk)
                    self.count_flops("f32", 2 * M * Cin * Cp)
                    if mk is not None:
                        masked.add(key)
                elif upsampled:
                    # A strided 3x3 (the last unit of a block, resnet_v1.py:80-113): dX = the stride-1 convolution of dY spread out on the
                    # input grid (zeros between its pixels) with the flipped / transposed filter -- the matrix pipe on a 75 % empty operand
                    # (~45 us) instead of the gather kernel's scalar FMAs (246 us, profiles/r04_ac_train_kernels_by_shape.txt).  The
                    # buffer is zeroed once: only the pixels (oh * stride, ow * stride) are ever written.
                    wd = prepared(("wflip", sc), lambda wf=wf, sc=sc, k=k, Cout=Cout: ops.flip_transpose_filter(
                        wf, out=sess.buf("bwd/wflip/" + sc, (wf.shape[3], k, k, Cout))))
                    up = sess.buf("bwd/up/" + sc, (N, up_h, up_w, Cout), zero=True)
                    ops.add_strided(gy, up, stride, False)
                    ops.conv2d(up, wd, None, k, k, 1, up_pad, ACT_NONE, gres if had else None, 1, out=gx, mask=mk)
                    self.count_flops("f32", 2 * x.shape[0] * x.shape[1] * x.shape[2] * Cin * Cout * k * k)
                    if mk is not None:
                        masked.add(key)
                else:
                    ops.conv2d_dgrad_strided(gy, wf, stride, pad, x.shape[1], x.shape[2], gx, had)
                    self.count_flops("f32", 2 * M * Cout * k * k * wf.shape[3])
            emitted.pop(y.data_ptr(), None)
            if solver is not None and p is not None:
                pending[0] += 1
                first = self._sgd_entry.get(sc)
                if dp:
                    if pending[0] >= self.SOLVER_CHUNK:
                        dp_solver_step()                         # (what the exchange has taken so far; pending counts on if nothing new was sent)
                elif pending[0] >= self.SOLVER_CHUNK and first is not None and first < self._sgd_done_from:
                    solver_step(first)                           # this record's data gradient is enqueued: its filter has no reader left
        for side in sides:
            ops.st_wait_stream(main, side)       # the solver (and the next forward pass, which overwrites x) come after every wgrad
        if solver is not None:
            ops.st_wait_stream(main, solver)
        return grads

    SOLVER_CHUNK = 32

    def _solver_stream_obj(self):
        if getattr(self, "_solver_stream", None) is None:
            self._solver_stream = torch.cuda.Stream(device=self.sess.device)
            self._solver_events = [torch.cuda.Event() for _ in range(8)]
        return self._solver_stream

    def _reg_total(self):
        if getattr(self, "_reg_buf", None) is None:
            self._reg_buf = torch.zeros((1,), dtype=torch.float32, device=self.sess.device)
        return self._reg_buf

    def regularization_value(self):
        """Device tensor [1]: the slim L2 term of this step's weights (network.py:315-317) -- computed by the sweep on the solver stream
        when that is active (before its first update), else here."""
        if not getattr(self, "_reg_in_sweep", False):
            self.regularization_loss(self._reg_total())
        self._reg_in_sweep = False
        return self._reg_total()

    def _wgrad_side_streams(self, n):
        have = getattr(self, "_wgrad_stream_objs", None)
        if have is None:
            have = self._wgrad_stream_objs = []
        while len(have) < n:
            have.append(torch.cuda.Stream(device=self.sess.device))
        return have[:n]

    # ---- solver --------------------------------------------------------------------------------------
    def apply(self, lr, world_size=1, all_reduce=None):
        """acc = m*acc + g ; w -= lr*acc (train_val.py:128-145) for every parameter in ONE launch.  Data parallel: the flat
        gradient is summed over the ranks (RCCL) -- bucket by bucket during the reverse sweep when `all_reduce` has the
        bucketed interface (parallel.BucketedAllReduce: backward() hands it every finished range), otherwise in one call
        here -- and the mean over replicas is folded into the SGD kernel (grad_scale = 1 / world_size)."""
        if all_reduce is not None and (world_size > 1 or self.data_parallel()):
            if hasattr(all_reduce, "finish"):
                all_reduce.finish(self.flat)              # ranges not yet handed over + wait for the ones in flight
            else:
                all_reduce(self.flat)
        gs = 1.0 / float(world_size)
        self.sess.prepared.join()                    # (a step that used none of the side stream's buffers has not waited for it yet)
        if getattr(self, "_sgd_table", None) is None:
            entries = []
            for p in self.params.values():
                if getattr(p, "dw", False):          # gradient already carries the BN-fold scale; no L2 term (MOBILENET.REGU_DEPTH False)
                    entries.append((p.w, p.acc_w, None, p.grad_w, None, p.K, 1.0, 0.0))
                    continue
                wd = self._wd(p.scope)
                entries.append((p.w, p.acc_w, p.wf if p.scale is not None else None, p.grad_w, p.scale, p.K, 1.0, wd))
                if p.bias is not None:
                    entries.append((p.bias, p.acc_b, None, p.grad_b, None, p.bias.numel(), 2.0 if self.double_bias else 1.0,
                                    wd if self.bias_decay else 0.0))
            self._sgd_count = len(entries)
            self._sgd_table = ops.sgd_desc_table(entries, self.sess.device)
            # gradient offset of every descriptor inside the flat buffer (ascending: the table is in forward order like the buffer)
            self._sgd_offsets = None if self.flat is None else [(e[3].data_ptr() - self.flat.data_ptr()) // 4 for e in entries]
            assert self._sgd_offsets is None or self._sgd_offsets == sorted(self._sgd_offsets)
            self._sgd_entry, i = {}, 0                 # scope -> index of its first descriptor (filter, then bias)
            for p in self.params.values():
                self._sgd_entry[p.scope] = i
                i += 1 if (getattr(p, "dw", False) or p.bias is None) else 2
        left = self._sgd_count if getattr(self, "_sgd_done_from", None) is None else self._sgd_done_from      # (the sweep updated the rest)
        if left != self._sgd_count and float(lr) != float(self.lr):
            raise RuntimeError("TrainState.apply(lr=%r): the reverse sweep already updated part of the parameters with TrainState.lr = %r"
                               % (lr, self.lr))
        self._sgd_done_from = None
        if left > 0:
            ops.sgd_momentum_range(self._sgd_table, 0, left, lr, self.momentum, gs)
        self.refresh_derived()

    def refresh_derived(self):
        """Everything computed FROM the trainable filters, after they changed (the solver's update; a caller that wrote Param.w / .wf
        itself): folded depthwise copies, then the derived filter images -- the operand planes of the forward pass's frcnn_gemm_h2
        launches and what a TEST-mode network on the same session reads: Winograd U of the 3x3 filters first (also with x3 / h2 off: a
        no-op when no ('wino', ...) entry is cached), then the pre-split planes of everything (x3 / h2) -- and the prepared gradient
        filters.  Weight-only launches: with cfg.HIP.PREP_STREAM they run on the side stream, beside the next forward pass
        (Session.h2_planes & co. wait for them at their first use: PreparedFilters.wait_planes)."""
        self.sess.device_filters_moved = True        # (Session.winograd_params: later first uses derive from the live device filters)
        for p in self.params.values():
            if getattr(p, "dw", False):
                ops.dwconv3x3_refold(p.w, p.scale, p.wf)

        def derived():
            self.sess.wino_refresh()
            if self.sess.x3:
                self.sess.x3_refresh()
            if self.sess.h2:                         # the same for cfg.HIP.MFMA_H2
                self.sess.h2_refresh()
        self.sess.prepared.weights_changed()
        self.sess.prepared.refresh(pre=derived)

    def invalidate_prepared(self):
        self.sess.prepared.invalidate()

    def _wd(self, scope):
        wd = self.net.weight_decay_for(scope) if hasattr(self.net, "weight_decay_for") else None
        return self.weight_decay if wd is None else wd

    # ---- checkpoint view (tf.train.Saver saves the variables AND the optimizer slots `<variable>/Momentum`) ----------
    def _names(self, p):
        return p.scope + "/weights", p.scope + "/biases"

    def export_variables(self, slots=True):
        """{TF/slim variable name: ndarray in the variable's own layout (HWIO filters, [in,out] matrices)} of every TRAINED
        parameter, read back from the device master copies (packed [Cout][KH][KW][Cin]); with slots=True also the momentum
        accumulators under `<name>/Momentum` (MomentumOptimizer's slot name)."""
        out = {}
        for p in self.params.values():
            if getattr(p, "dw", False):
                name = p.scope + "/depthwise_weights"
                for suffix, t in (("", p.w),) + ((("/Momentum", p.acc_w),) if slots else ()):
                    out[name + suffix] = t.detach().cpu().numpy().reshape(self.sess.variables[name].shape).copy()
                continue
            wname, bname = self._names(p)
            ref = self.sess.variables[wname]
            for suffix, t in (("", p.w),) + ((("/Momentum", p.acc_w),) if slots else ()):
                hwio = t.detach().cpu().numpy().transpose(1, 2, 3, 0)          # [O,KH,KW,I] -> [KH,KW,I,O]
                if hwio.size != ref.size:
                    raise ValueError("%s: device filter has %d values, variable %d (channel-folded stem filters are not exportable)"
                                     % (wname, hwio.size, ref.size))
                out[wname + suffix] = np.ascontiguousarray(hwio).reshape(ref.shape)
            if p.bias is not None:
                out[bname] = p.bias.detach().cpu().numpy().copy()
                if slots:
                    out[bname + "/Momentum"] = p.acc_b.detach().cpu().numpy().copy()
        return out

    def import_slots(self, reader_or_dict):
        """Restore the momentum accumulators written by export_variables(slots=True)."""
        get = reader_or_dict.get_tensor if hasattr(reader_or_dict, "get_tensor") else reader_or_dict.__getitem__
        for p in self.params.values():
            if getattr(p, "dw", False):
                acc = np.asarray(get(p.scope + "/depthwise_weights/Momentum"), dtype=np.float32)
                p.acc_w.copy_(torch.from_numpy(np.ascontiguousarray(acc.reshape(tuple(p.w.shape)))))
                continue
            wname, bname = self._names(p)
            acc = np.asarray(get(wname + "/Momentum"), dtype=np.float32)
            kh, kw = p.w.shape[1], p.w.shape[2]
            acc = acc.reshape(kh, kw, p.w.shape[3], p.w.shape[0]).transpose(3, 0, 1, 2)
            p.acc_w.copy_(torch.from_numpy(np.ascontiguousarray(acc)))
            if p.bias is not None:
                p.acc_b.copy_(torch.from_numpy(np.asarray(get(bname + "/Momentum"), dtype=np.float32)))

    def regularization_loss(self, out):
        """slim l2_regularizer(WEIGHT_DECAY): wd * sum(w^2)/2 over every conv / fc weight (network.py:315-317), all tensors in
        two launches (the master tensors are static, so the pointer table is built once)."""
        if getattr(self, "_reg_tables", None) is None:
            groups = {}                                              # L2 coefficient -> tensors (MobileNet: backbone vs heads)
            for sc in self.reg_scopes:
                p = self.params.get(sc)
                if p is not None:
                    t = p.w
                else:
                    info = self.sess.conv_info[sc]
                    if "w_master" not in info:
                        info["w_master"] = info["w"] if info["scale"] is None else \
                            (info["w"] / torch.from_numpy(info["scale"]).to(info["w"].device).view(-1, 1, 1, 1)).contiguous()
                    t = info["w_master"]
                groups.setdefault(self._wd(sc), []).append(t)
            dev = self.sess.device
            self._reg_keep = groups
            self._reg_tables = [(wd, torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev),
                                 torch.tensor([t.numel() for t in ts], dtype=torch.int64, device=dev)) for wd, ts in groups.items()]
        for i, (wd, ptrs, sizes) in enumerate(self._reg_tables):
            ops.sumsq_multi(ptrs, sizes, 0.5 * wd, out, i > 0)
        return out
