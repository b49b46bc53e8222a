"""Session -- the opaque runtime handle that replaces `tf.Session` in the host-side mirror
(SURVEY.md 8b "Model-level seam"): device + stream + variables (host numpy, TF layouts and names)
+ their packed device images + static activation buffers + captured hipGraphs.

torch is used for device memory and streams only.
"""
import collections
import os
import time

import numpy as np
import torch

from . import ops


class VarSpec(object):
    __slots__ = ("shape", "init", "arg")

    def __init__(self, shape, init, arg=None):
        self.shape, self.init, self.arg = tuple(int(s) for s in shape), init, arg


class VariableStore(object):
    """Host side of a session: the variables under their TF/slim names (numpy, HWIO filters, [in,out] matrices), their
    initialisation and checkpoint I/O.  No device state -- Session adds that."""

    def __init__(self, seed=3):
        self.variables = collections.OrderedDict()     # TF name -> numpy (HWIO conv, [in,out] fc)
        self.packed = {}                                # layer key -> device tensors
        self.x3 = {}                                    # (filter address, shape) -> (bf16 planes, filter): frcnn_gemm_x3 operands
        self.h2 = {}                                    # (filter address, shape) -> ((fp16 planes, w_inv), filter): frcnn_gemm_h2 operands
        self.h2_spread = {}                             # scope -> h2_channel_spread (host statistic of the filter)
        self.conv_info = {}                             # scope -> {w (folded, device), b, scale (np or None), bn}
        self.graphs = {}
        self.seed = seed
        self.device_filters_moved = False               # a solver updated the device filters in place: the host variables are older (TrainState)
        self.derived_gen = 0                            # bumped whenever a derived filter image is ADDED or the set is dropped (derived_generation)

    # ---- variables ---------------------------------------------------------------------------
    def init_variables(self, specs, seed=None):
        """Random-init weights of the reference architecture (no checkpoints exist offline):
        He/variance-scaling for backbone convs (lib/nets/resnet_v1.py:38), N(0,0.01)/N(0,0.001)
        for RPN/cls/bbox (lib/nets/network.py:239-240), zero biases (:420), synthetic frozen-BN
        statistics."""
        rng = np.random.RandomState(self.seed if seed is None else seed)
        truncated = False
        try:
            from model.config import cfg
            truncated = bool(cfg.TRAIN.TRUNCATED)
        except ImportError:
            pass
        def randn_scaled(shape, scale):
            """(rng.randn(*shape) * scale).astype(float32), drawn in 2^20-sample pieces: the same stream and the same float64 product, without
            the two float64 temporaries of the whole tensor (VGG16's fc6 is 103 M elements: 1.6 GB of first-touch pages per network)."""
            n = int(np.prod(shape))
            out = np.empty((n,), dtype=np.float32)
            for o in range(0, n, 1 << 20):
                m = min(1 << 20, n - o)
                out[o:o + m] = rng.randn(m) * scale
            return out.reshape(shape)

        for name, sp in specs.items():
            if sp.init == "he":
                fan_in = int(np.prod(sp.shape[:-1]))
                v = randn_scaled(sp.shape, np.sqrt(2.0 / fan_in))
            elif sp.init == "normal" and not truncated:
                v = randn_scaled(sp.shape, sp.arg)
            elif sp.init == "normal":
                v = rng.randn(*sp.shape)      # cfg.TRAIN.TRUNCATED (network.py:235-240): tf.truncated_normal_initializer re-draws |z| > 2
                bad = np.abs(v) > 2.0
                while bad.any():
                    v[bad] = rng.randn(int(bad.sum()))
                    bad = np.abs(v) > 2.0
                v = v * sp.arg
            elif sp.init == "zeros":
                v = np.zeros(sp.shape)
            elif sp.init == "bn_gamma":
                v = rng.uniform(0.5, 1.5, size=sp.shape)
            elif sp.init == "bn_gamma_res":
                # last BN of a residual branch: small gain, otherwise 33 stacked units with
                # un-matched synthetic statistics double the activation variance per unit
                v = rng.uniform(0.1, 0.3, size=sp.shape)
            elif sp.init == "bn_var":
                v = rng.uniform(0.5, 1.5, size=sp.shape)
            elif sp.init == "bn_beta" or sp.init == "bn_mean":
                v = rng.randn(*sp.shape) * 0.1
            else:
                raise ValueError(sp.init)
            self.variables[name] = v.astype(np.float32)
        self.conv_info.clear()
        self._drop_derived()

    def _drop_derived(self):
        """Everything computed FROM the variables: packed device images, pre-split x3 / h2 filter planes, captured graphs and (Session)
        the prepared-filter plan whose closures hold the filter tensors -- a direct load / restore between training steps must not leave
        the data-gradient chain reading flipped / Winograd / h2 filters derived from tensors that were just dropped."""
        self.packed.clear()
        self.x3.clear()
        self.h2.clear()
        self.h2_spread.clear()
        self.graphs.clear()
        self.derived_gen += 1
        self.device_filters_moved = False               # (the device images are rebuilt from the host variables from here on)
        prepared = getattr(self, "prepared", None)
        if prepared is not None:
            prepared.invalidate()

    def load_variables(self, values):
        for k, v in values.items():
            self.variables[k] = np.asarray(v, dtype=np.float32)
        self._drop_derived()

    def restore(self, ckpt_prefix, names=None, verify=True):
        """tf.train.Saver(...).restore(sess, ckpt) without TensorFlow (tools/test_net.py:110-114, train_val.py:185-190):
        loads `names` (default: every declared variable) from a V2 checkpoint by their TF/slim names.  A missing name or a
        shape mismatch raises, like Saver does.  Returns the names restored."""
        from .tensor_bundle import open_checkpoint
        reader = open_checkpoint(ckpt_prefix, verify=verify)          # V2 bundle or V1 single file
        names = list(self.variables) if names is None else list(names)
        values = {}
        for name in names:
            if not reader.has_tensor(name):
                raise KeyError("Key %s not found in checkpoint %s" % (name, ckpt_prefix))
            v = reader.get_tensor(name)
            if name in self.variables and tuple(v.shape) != tuple(self.variables[name].shape):
                if v.size != self.variables[name].size or name.rsplit("/", 2)[-2] not in ("fc6", "fc7"):
                    raise ValueError("%s: checkpoint shape %s, variable shape %s" % (name, list(v.shape), list(self.variables[name].shape)))
                v = v.reshape(self.variables[name].shape)     # the fc6 / fc7 matrix stored as a filter or vice versa (vgg16.py:81-100)
            values[name] = v
        self.load_variables(values)
        self.conv_info.clear()
        return names

    def save(self, ckpt_prefix, extra=None):
        """tf.train.Saver.save: every variable (+ `extra`, e.g. momentum slots / global_step) as a V2 checkpoint."""
        from .tensor_bundle import write_bundle
        tensors = {k: np.asarray(v) for k, v in self.variables.items()}
        tensors.update(extra or {})
        return write_bundle(ckpt_prefix, tensors)


class PreparedFilters(object):
    """Weight-only launches of a TRAINING step -- the Winograd transforms of the 3x3 filters (forward and gradient filter), the flipped /
    transposed filters of the data gradients and their h2 split, the operand planes of the forward pass's filters: ~200 tiny kernels that
    depend on nothing but the filters, yet sit on the chain every activation waits for.  Each call site hands its launches over as
    `get(key, fn)`; fn() enqueues them into buffers of its own and returns those.  The first time (and whenever the filters changed
    behind the solver's back) fn runs inline; the solver calls weights_changed() + refresh() right after its update, which re-runs every
    remembered fn on a side stream -- beside the next forward pass -- and the next get() finds the buffers done.  Three tiers, one event
    each, waited for once per step at the first use: 0 = the session's cached filter images (refresh(pre=...): operand planes, TEST-mode
    Winograd U; readers call wait_planes()), 1 = the forward pass's entries (keys ("fwd", ...)), 2 = the reverse sweep's."""

    def __init__(self, device):
        self.device = device
        self.enabled = True
        self.plan = {}                          # key -> (fn, buffers), insertion order = first-use order (a flip before its split)
        self.ready = frozenset()
        self.version, self.ready_version = 0, -1
        self.stream = None
        self.events = None                      # one per tier
        self.epoch = 0                          # counts refreshes (incl. replayed ones: refreshed())
        self.gen = 0                            # bumped when a key is added to the plan / the plan is dropped (Session.derived_generation)
        self.waited = {}                        # stream handle -> [epoch whose tier-t event this stream has waited for, t = 0, 1, 2]
        self.readers = {}                       # stream handle -> torch stream: who has read the buffers since the last refresh

    def _wait(self, tier):
        """Per STREAM: a TEST-mode network on its own stream and the training stream each wait for the refresh once (one shared flag let
        the second reader skip its wait).  Tiers complete in order on the refresh stream: waiting for one covers the earlier ones."""
        cur = torch.cuda.current_stream(self.device)
        h = int(cur.cuda_stream)
        w = self.waited.get(h)
        if w is None:
            w = self.waited[h] = [0, 0, 0]
        self.readers[h] = cur
        if w[tier] < self.epoch:
            ops.st_wait_event(cur, self.events[tier])
            for t in range(tier + 1):
                w[t] = self.epoch

    def refreshed(self):
        """the tier events were re-recorded by somebody else than refresh() (a replayed training step's refresh): every reader waits again"""
        self.epoch += 1

    def get(self, key, fn):
        if self.enabled and self.ready_version == self.version and key in self.ready:
            self._wait(1 if key[0] == "fwd" else 2)     # the forward pass does not wait for the gradient filters queued behind its own
            return self.plan[key][1]
        with ops.unscoped():                # filter images are per session, not per image shape (Session.shape_scope)
            out = fn()
        if self.enabled:
            if key not in self.plan:
                self.gen += 1
            self.plan[key] = (fn, out)
        return out

    def forget_waits(self, stream):
        """A recording of a training step starts on `stream`: whatever this stream has already waited for this epoch (a TEST-mode forward or
        wait_planes() between two steps), the RECORDED step must carry the wait of every tier at its first use -- a replay re-records the
        tier events and its forward pass has to order itself behind them (a recording made after an earlier tier-0 wait had
        no wait on events[0])."""
        self.waited.pop(int(stream.cuda_stream), None)

    def wait_planes(self):
        """Before a launch reads a cached filter image of the session (Session.h2_planes / x3_planes / winograd_params, and a TEST-mode
        graph replay, whose launches read them without passing through those accessors)."""
        if self.events is not None:
            self._wait(0)

    def join(self):
        """Before the filters are written again (the solver's update): everything the last refresh() enqueued has read them."""
        if self.events is not None:
            self._wait(2)

    def weights_changed(self):
        self.version += 1

    def refresh(self, pre=None):
        """pre: launches that re-derive the session's cached filter images from the updated filters (tier 0)."""
        if not self.enabled:
            self.ready = frozenset()
            if pre is not None:
                pre()
            return
        main = torch.cuda.current_stream(self.device)
        if self.stream is None:
            self.stream, self.events = torch.cuda.Stream(device=self.device), [torch.cuda.Event() for _ in range(3)]
        ops.st_wait_stream(self.stream, main)   # the update itself, and the last step's reads of these buffers
        for h, st in list(self.readers.items()):   # ... and whoever else read them (a TEST-mode network on another stream)
            if h != int(main.cuda_stream):
                ops.st_wait_stream(self.stream, st)
        self.readers = {}
        with ops.pinned_stream(self.stream), ops.unscoped():
            if pre is not None:
                pre()
            ops.ev_record(self.events[0], self.stream)
            for key, (fn, _) in self.plan.items():
                if key[0] == "fwd":
                    fn()
            ops.ev_record(self.events[1], self.stream)
            for key, (fn, _) in self.plan.items():
                if key[0] != "fwd":
                    fn()
        ops.ev_record(self.events[2], self.stream)
        self.ready, self.ready_version = frozenset(self.plan), self.version
        self.epoch += 1

    def invalidate(self):
        """The filter tensors were replaced or rewritten by somebody else than the solver (restore, initialise): forget the plan (its
        closures hold the tensors); the next step prepares inline again."""
        self.version += 1
        self.gen += 1
        self.plan, self.ready = {}, frozenset()


class Session(VariableStore):
    def __init__(self, device=None, seed=3):
        if not torch.cuda.is_available():
            raise RuntimeError("frcnn_hip.Session needs a GPU: the product path has no CPU fallback")
        VariableStore.__init__(self, seed)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.buffers = {}                               # session-wide buffers (weight-shaped, solver state, anything allocated outside a scope)
        self.scopes = collections.OrderedDict()         # shape-scope key -> {buffer key: tensor}, least recently entered first (shape_scope)
        self.scope_group = {}                           # shape-scope key -> LRU group
        # debugging aids (tests; FRCNN_SCOPE_POISON=1 / FRCNN_SCOPE_CAP=n in the environment): new buffers start as NaN patterns instead of
        # whatever the allocator hands out, so a launch that reads memory nobody wrote shows up at once; a cap for every LRU group
        self.poison_new_buffers = os.environ.get("FRCNN_SCOPE_POISON", "") not in ("", "0")
        self.scope_cap_override = int(os.environ.get("FRCNN_SCOPE_CAP", "0")) or None
        self.profile = None                             # list of (tag, flops, ev0, ev1) when profiling
        self.flops_last_forward = 0
        self.flops_by_pipe = None                       # dict while somebody wants the split (Session.mark)
        self.prepared = PreparedFilters(self.device)

    # ---- device-side images of the variables ----------------------------------------------------
    def to_device(self, a, dtype=torch.float32):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, dtype)

    def fold_bn(self, scope, eps):
        v = self.variables
        g, b = v[scope + "/BatchNorm/gamma"], v[scope + "/BatchNorm/beta"]
        m, var = v[scope + "/BatchNorm/moving_mean"], v[scope + "/BatchNorm/moving_variance"]
        scale = (g.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps))
        bias = b.astype(np.float64) - m.astype(np.float64) * scale
        return scale.astype(np.float32), bias.astype(np.float32)

    def conv_params(self, scope, bn_eps=None, fold_w=False, out_scale=None, out_shift=None):
        """(w_packed_d, bias_d) for a conv / fc scope.  bn_eps: fold the frozen batch norm
        gamma*(x-mean)/sqrt(var+eps)+beta into (w*scale, bias).  out_scale/out_shift: extra affine
        on the outputs (test-time bbox de-normalisation, lib/nets/network.py:428-432)."""
        key = (scope, bn_eps, fold_w, None if out_scale is None else tuple(out_scale))
        if key in self.packed:
            return self.packed[key]
        w = self.variables[scope + "/weights"]
        if w.ndim == 2:
            w = w[None, None, :, :]
        scale, bias = None, None
        if bn_eps is not None:
            scale, bias = self.fold_bn(scope, bn_eps)
        elif (scope + "/biases") in self.variables:
            bias = self.variables[scope + "/biases"]
        if out_scale is not None:
            osc = np.asarray(out_scale, dtype=np.float32)
            scale = osc if scale is None else scale * osc
            bias = (np.zeros_like(osc) if bias is None else bias) * osc + np.asarray(out_shift, dtype=np.float32)
        wp = ops.pack_filter_foldw(w, scale) if fold_w else ops.pack_filter_hwio(w, scale)
        res = (self.to_device(wp), None if bias is None else self.to_device(bias))
        self.packed[key] = res
        self.conv_info[scope] = {"w": res[0], "b": res[1], "scale": scale if bn_eps is not None else None, "bn": bn_eps is not None}
        return res

    def winograd_params(self, scope, bn_eps=None, m=2):
        """(U_d [(m+2)^2,Cout,Cin], bias_d) for a 3x3 stride-1 scope run as Winograd F(m x m,3x3); m = 7: the mixed 7x7 scheme,
        U_d [121,Cout,Cin] (TEST mode; training transforms the live filter on the device)."""
        key = ("wino", scope, bn_eps, m)
        self.prepared.wait_planes()
        if key in self.packed:
            return self.packed[key]
        info = self.conv_info.get(scope)
        if self.device_filters_moved and info is not None and info["w"].dim() == 4 and tuple(info["w"].shape[1:3]) == (3, 3) and info["bn"] == (bn_eps is not None):
            # a solver on this session has updated the filters IN PLACE on the device since the host variables were loaded (TrainState.
            # refresh_derived): U comes from the live folded filter -- what wino_refresh() will re-derive it from after every later step
            # -- and the bias is the live tensor; the host copy is the pre-training filter
            res = (ops.winograd_filter_transform_device(info["w"], m, False), info["b"])
        else:
            w = self.variables[scope + "/weights"]
            scale, bias = None, None
            if bn_eps is not None:
                scale, bias = self.fold_bn(scope, bn_eps)
            elif (scope + "/biases") in self.variables:
                bias = self.variables[scope + "/biases"]
            res = (self.to_device(ops.winograd_filter_transform(w, scale, m)), None if bias is None else self.to_device(bias))
        self.packed[key] = res
        self.derived_gen += 1
        return res

    def derived_generation(self):
        """Identifies the SET of filter images derived from the variables: the session's cached operand planes / Winograd filters
        (h2, x3, ("wino", ...)) and the prepared-filter plan.  A recorded training step (lib/nets/network.py _train_step_replayed) re-derives
        exactly the set that existed when it was recorded; an entry added later -- a TEST-mode network first run on the shared session, another
        image shape's plan key -- would never be refreshed by its replays while ready / ready_version keep reporting it current.  The
        recording stores this value and is dropped when it no longer matches."""
        return (self.derived_gen, self.prepared.gen)

    # ---- buffers: per session, or per image shape -------------------------------------------------------------------------------------
    # The reference's graph takes [1, None, None, 3] and test_net walks an imdb whose images all differ in size (lib/nets/network.py:386-390,
    # lib/model/test.py:138-185).  Here every distinct shape has its own static buffers (a captured hipGraph / a recorded step addresses
    # them), so they are grouped per shape and the groups are kept least-recently-used: entering a scope beyond the group's cap drops the
    # oldest shapes -- captured graph, activation buffers, operand planes, arena results, scratch -- back to torch's allocator, whose
    # size-class pools hand the blocks to the next shape.  Memory is bounded by cap shapes per group, not by the imdb.
    class _Scope(object):
        def __init__(self, sess, key, group, cap):
            self.sess, self.key, self.group, self.cap = sess, key, group, cap

        def __enter__(self):
            s = self.sess
            if self.key not in s.scopes:
                if self.cap is not None:
                    cap = getattr(s, "scope_cap_override", None) or self.cap
                    live = [k for k in s.scopes if s.scope_group.get(k) == self.group]
                    drop = live[:max(0, len(live) + 1 - max(1, int(cap)))]
                    if drop:
                        torch.cuda.synchronize(s.device)          # a replay of an evicted graph may still be running
                        for k in drop:
                            s.drop_scope(k)
                s.scopes[self.key] = {}
                s.scope_group[self.key] = self.group
            s.scopes.move_to_end(self.key)
            self.prev, ops.scope_store = ops.scope_store, s.scopes[self.key]
            return self

        def __exit__(self, *exc):
            ops.scope_store = self.prev
            return False

    def shape_scope(self, key, group=None, cap=None):
        """`with sess.shape_scope(key, group, cap):` -- buffers requested inside belong to `key` (an image shape's graph key / recorded-step
        key).  At most `cap` scopes of a group stay alive; the least recently entered ones go first, with their sess.graphs entry."""
        return Session._Scope(self, key, group, cap)

    def drop_scope(self, key):
        """Forget one shape: its captured graph / recording (sess.graphs[key]) and every buffer registered under it.  The caller makes sure
        nothing that addresses them is still running (shape_scope synchronises before an eviction)."""
        for k, ent in list(self.graphs.items()):          # the shape's own graph, and every recorded step that addresses its buffers
            if k == key or (isinstance(ent, dict) and ent.get("scope") == key):
                pk = ent.get("picker") if isinstance(ent, dict) else None
                if pk is not None and not pk.done:
                    self.picking = False                  # (a stream search that was running on the dropped recording ends with it)
                del self.graphs[k]
        self.scope_group.pop(key, None)
        return self.scopes.pop(key, None) is not None

    def _store(self):
        return self.buffers if ops.scope_store is None else ops.scope_store

    def find_buf(self, name, shape, dtype=torch.float32):
        """A named buffer wherever it lives: the active scope, the most recently entered shape scopes, the session (tests / harnesses that
        inspect an intermediate tensor after a run; the product path asks through buf() inside the right scope)."""
        key = (name, tuple(shape), dtype)
        stores = ([ops.scope_store] if ops.scope_store is not None else []) + list(reversed(self.scopes.values())) + [self.buffers]
        for st in stores:
            if key in st:
                return st[key]
        raise KeyError(key)

    def scope_bytes(self, key=None):
        """bytes held by one shape scope (key) or by all of them: what an eviction returns"""
        def size(v):
            if torch.is_tensor(v):
                return v.numel() * v.element_size()
            if isinstance(v, (tuple, list)):
                return sum(size(x) for x in v)
            return sum(size(getattr(v, a)) for a in ("planes", "inv") if hasattr(v, a))
        stores = [self.scopes[key]] if key is not None else list(self.scopes.values())
        return sum(size(v) for st in stores for v in st.values())

    def buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        store = self._store()
        t = store.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            if getattr(self, "poison_new_buffers", False) and not zero:
                t.view(torch.uint8).fill_(0xFF)         # float32 / fp16 NaN, int32 -1
            store[key] = t
        return t

    # ---- profiling hook: HIP events around selected launches, on the stream they run on ----------
    def x3_planes(self, w):
        """Pre-split bf16 planes of a static device filter for frcnn_gemm_x3 (cfg.HIP.MFMA_X3), split once and cached for the
        life of the session (the entry keeps the filter alive: its address is the key)."""
        key = (w.data_ptr(), tuple(w.shape))
        self.prepared.wait_planes()
        ent = self.x3.get(key)
        if ent is None:
            ent = (ops.gemm_x3_pack(w), w)
            self.x3[key] = ent
            self.derived_gen += 1
        return ent[0]

    def wino_refresh(self):
        """Recompute every cached TEST-mode Winograd filter U (winograd_params) from the LIVE folded device filter, into its existing
        buffer: the solver updates filters in place, and a TEST-mode network sharing the session must not keep multiplying by the
        transform of the pre-training 3x3 filters (x3_refresh / h2_refresh then re-split U like any other filter).  Returns the count."""
        n = 0
        for key, val in list(self.packed.items()):
            if isinstance(key, tuple) and key and key[0] == "wino":
                u = val[0]
                info = self.conv_info.get(key[1])
                if info is not None and info["w"].dim() == 4 and info["w"].shape[1] == 3:
                    ops.winograd_filter_transform_device(info["w"], key[3], False, out=u)
                    n += 1
        return n

    def x3_refresh(self):
        """Re-split every cached filter into its EXISTING plane buffer (addresses captured by hipGraphs stay valid): the solver
        calls this after updating filters in place, so a TEST-mode network on the same session never multiplies by stale planes."""
        for planes, w in self.x3.values():
            ops.gemm_x3_pack(w, planes)

    def h2_channel_spread(self, scope):
        """min over input channels k of (largest |folded filter entry| that multiplies channel k) / (largest entry of the filter): host
        arithmetic on the variables, once per scope.  A trained network with outlier activation channels (1e3 ... 1e5 times the rest)
        carries correspondingly small filter entries for them in the consumer; below 2^-18 the block-scaled fp16x2 format of
        frcnn_gemm_h2 would hold those entries (and the non-outlier activations sharing a 128-k block with the outlier) to fewer bits
        than float32 does, so such a layer stays on the exact x3 split (lib/nets/network.py _h2_eligible)."""
        got = self.h2_spread.get(scope)
        if got is None:
            w = np.abs(self.variables[scope + "/weights"].astype(np.float64))
            if w.ndim == 2:
                w = w[None, None]
            if (scope + "/BatchNorm/gamma") in self.variables:
                v = self.variables
                w = w * np.abs(v[scope + "/BatchNorm/gamma"].astype(np.float64) / np.sqrt(v[scope + "/BatchNorm/moving_variance"].astype(np.float64) + 1e-5))
            per_k = w.max(axis=(0, 1, 3))
            top = float(per_k.max())
            nz = per_k[per_k > 0]
            got = self.h2_spread[scope] = (float(nz.min()) / top) if (top > 0 and nz.size) else 1.0
        return got

    def h2_planes(self, w):
        """Pre-split fp16 planes + per-row scales of a static device filter [N, ...K] / [G, N, K] for frcnn_gemm_h2 (cfg.HIP.MFMA_H2),
        split once and cached like x3_planes."""
        key = (w.data_ptr(), tuple(w.shape))
        self.prepared.wait_planes()                 # (training: the solver re-splits the cached filters on its side stream)
        ent = self.h2.get(key)
        if ent is None:
            ent = (ops.h2_pack_w(w), w)
            self.h2[key] = ent
            self.derived_gen += 1
        return ent[0]

    def h2_refresh(self):
        """Re-split every cached filter into its existing buffers (see x3_refresh)."""
        for packed, w in self.h2.values():
            ops.h2_pack_w(w, packed)

    def h2_buf(self, name, rows, K):
        """Static operand-plane buffer (ops.H2) for an activation tensor of [rows, K]."""
        key = ("h2", name, int(rows), int(K))
        store = self._store()
        t = store.get(key)
        if t is None:
            t = store[key] = ops.H2.empty(rows, K, self.device)
        return t

    def buf_pair(self, name, N, K):
        """Static (planes, w_inv) pair for ops.h2_pack_w(out=...) of a [N, K] filter that changes every step (training)."""
        key = ("h2w", name, int(N), int(K))
        t = self.buffers.get(key)
        if t is None:
            t = self.buffers[key] = (torch.empty(int(ops.lib().frcnn_h2_planes_bytes(int(N), int(K))), dtype=torch.uint8, device=self.device),
                                     torch.empty((1, int(N)), dtype=torch.float32, device=self.device))
        return t

    def mark(self, tag, flops, fn, nbytes=0):
        """nbytes: algorithmic HBM bytes of the launch (operands read once + result written once), for the roofline report."""
        self.flops_last_forward += flops
        if flops and self.flops_by_pipe is not None:        # bench.py: a forward pass's matrix work per matrix pipe (host bookkeeping)
            pipe = "h2" if tag.startswith("conv:h2:") else "x3" if tag.startswith("conv:x3:") else "f32"
            self.flops_by_pipe[pipe] = self.flops_by_pipe.get(pipe, 0) + flops
        if self.profile is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.profile.append((tag, flops, e0, e1, nbytes))
        return r

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        self.graphs.clear()
        self.scopes.clear()
        self.scope_group.clear()
        self.buffers.clear()
        self.packed.clear()
        self.x3.clear()
        self.h2.clear()


class Timer(object):
    """tic/toc wall-clock timer with the interface of lib/utils/timer.py:10-32."""

    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
