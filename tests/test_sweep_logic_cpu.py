"""CPU: the bookkeeping of the reverse sweep (frcnn_hip/train.py TrainState._sweep) -- which gradient buffer is masked in the producing
launch, which identity-shortcut gradient is borrowed instead of copied, where operand planes of dY are emitted, when a tensor that
received an unmasked contribution still gets its relu_bwd pass -- run against plain torch-CPU stand-ins for the C-ABI entries and for
the stream / event objects.  The stand-ins compute the same functions the entries document (a convolution, y = mask > 0 ? y : 0, ...),
so the sweep with the folded passes must give the SAME bits as the sweep with the separate passes, and both must agree with torch
autograd of the same small network (two bottleneck units with identity / projection shortcuts, a strided 3x3, odd-width heads, a tensor
with two consumers).  No GPU, no library: this is the host logic only; the kernels themselves are checked in tests/test_chain_fusion_gpu.py."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd"))

ACT_NONE, ACT_RELU = 0, 1


def nchw(t):
    return t.permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def w_oihw(w):                      # packed [Cout, KH, KW, Cin] -> torch [Cout, Cin, KH, KW]
    return w.permute(0, 3, 1, 2)


class FakeStream(object):
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, other):
        pass


class FakeEvent(object):
    def record(self, stream=None):
        pass


class Planes(object):
    """stand-in for ops.H2: the tensor itself"""

    def __init__(self, rows, K):
        self.rows, self.K, self.t = rows, K, torch.zeros((rows, K))


class FakeOps(types.SimpleNamespace):
    """torch-CPU versions of the entries the sweep calls; `log` counts the calls per entry"""

    def __init__(self):
        types.SimpleNamespace.__init__(self)
        self.log, self.ws_scope, self.filters = {}, "", {}

    def _n(self, name):
        self.log[name] = self.log.get(name, 0) + 1

    # the stream-level helpers of frcnn_hip.ops (recorded when a step is being recorded; here: the plain operations)
    @staticmethod
    def ev_record(ev, stream):
        ev.record(stream)

    @staticmethod
    def st_wait_event(stream, ev):
        stream.wait_event(ev)

    @staticmethod
    def st_wait_stream(stream, other):
        stream.wait_stream(other)

    @staticmethod
    def t_copy(dst, src):
        return dst.copy_(src)

    @staticmethod
    def t_zero(t):
        return t.zero_()

    class pinned_stream(object):
        def __init__(self, stream):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    # ---- elementwise
    def relu_bwd(self, g, y):
        self._n("relu_bwd")
        g.copy_(torch.where(y > 0, g, torch.zeros_like(g)))
        return g

    def add_strided(self, src, dst, stride, accumulate):
        self._n("add_strided")
        view = dst[:, ::stride, ::stride, :][:, :src.shape[1], :src.shape[2], :]
        view.copy_(view + src if accumulate else src)
        return dst

    def spatial_mean_bwd(self, dy, HW, out):
        self._n("spatial_mean_bwd")
        out.copy_((dy / float(HW)).view(dy.shape[0], 1, 1, dy.shape[1]).expand_as(out))
        return out

    def colsum(self, gy2d, out):
        out.copy_(gy2d.double().sum(0).float())

    def transpose_pad(self, x2d, Mp, out=None):
        out.zero_()
        out[:, :x2d.shape[0]] = x2d.t()
        return out

    # ---- convolutions
    @staticmethod
    def _mask(v, mask):
        return v if mask is None else torch.where(mask.reshape(v.shape) > 0, v, torch.zeros_like(v))

    def conv2d(self, x, w, bias, KH, KW, stride=1, pad=(0, 0, 0, 0), act=0, residual=None, res_stride=1, fold_w=False, out=None, mask=None):
        self._n("conv2d_masked" if mask is not None else "conv2d")
        xp = F.pad(nchw(x), (pad[2], pad[3], pad[0], pad[1]))
        v = nhwc(F.conv2d(xp, w_oihw(w.view(w.shape[0], KH, KW, -1)), stride=stride))
        if residual is not None:
            v = v + residual.reshape(v.shape)
        out.copy_(self._mask(v, mask).reshape(out.shape))
        return out

    def flip_transpose_filter(self, wf, out=None):
        out.copy_(wf.flip(1, 2).permute(3, 1, 2, 0))          # [Cout,k,k,Cin] -> [Cin,k,k,Cout], taps reversed
        return out

    def conv2d_dgrad_strided(self, gy, wf, stride, pad, H, W, gx, accumulate):
        self._n("dgrad_strided")
        full = F.conv_transpose2d(nchw(gy), w_oihw(wf), stride=stride)
        full = F.pad(full, (0, max(0, W + pad[2] - full.shape[3]), 0, max(0, H + pad[0] - full.shape[2])))
        v = nhwc(full[:, :, pad[0]:pad[0] + H, pad[2]:pad[2] + W])
        gx.copy_(gx + v if accumulate else v)
        return gx

    def conv2d_wgrad_supported(self, Cin, Cout):
        return True

    def conv2d_wgrad(self, gy, x, KH, KW, stride, pad, out, h2=False):
        self._n("wgrad")
        xp = F.pad(nchw(x), (pad[2], pad[3], pad[0], pad[1])).double()
        g = torch.nn.grad.conv2d_weight(xp, (gy.shape[3], x.shape[3], KH, KW), nchw(gy).double(), stride=stride)
        out.copy_(g.permute(0, 2, 3, 1).float().reshape(out.shape))
        return out

    # ---- Winograd path: the "transform" remembers the filter, the "convolution" is the plain one
    def winograd_points(self, m):
        return 121 if m == 7 else (m + 2) ** 2

    def winograd_tiles(self, N, H, W, m):
        return N if m == 7 else N * ((H + m - 1) // m) * ((W + m - 1) // m)

    def winograd_filter_transform_device(self, wf, m, transpose_flip, out=None):
        assert transpose_flip
        self.filters[out.data_ptr()] = wf
        return out

    def conv3x3_winograd(self, x, u, bias, act=0, out=None, v_buf=None, m_buf=None, u_planes=None, v_planes=None, mask=None, out_planes=None):
        self._n("wino_masked" if mask is not None else "wino")
        wf = self.filters[u.data_ptr()]
        v = nhwc(F.conv_transpose2d(nchw(x), w_oihw(wf), padding=1))
        v = self._mask(v, mask)
        out.copy_(v)
        if out_planes is not None:
            self._n("planes_from_wino")
            out_planes.t.copy_(v.reshape(out_planes.rows, out_planes.K))
        return out

    # ---- solver (train_val.py:128-145 semantics, as csrc/backward_kernels.hip k_sgd_multi states them)
    def sgd_desc_table(self, entries, device):
        return list(entries)

    def sgd_momentum_range(self, table, first, count, lr, momentum, grad_scale=1.0):
        self._n("sgd_range")
        assert 0 <= first and first + count <= len(table) and count > 0
        for w, acc, wf, grad, scale, K, lr_mult, wd in table[first:first + count]:
            sc = 1.0 if scale is None else scale.view(-1, 1, 1, 1)
            g = grad_scale * grad * sc + wd * w
            acc.copy_(momentum * acc + g)
            w.copy_(w - lr * lr_mult * acc)
            self.updated = getattr(self, "updated", [])
            self.updated.append(w.data_ptr())

    def sumsq_multi(self, ptrs, sizes, scale, out, accumulate=False):
        return out

    # ---- the fp16-pipe GEMM
    def h2_split(self, x2d, out=None):
        self._n("h2_split")
        out.t.copy_(x2d)
        return out

    def h2_pack_w(self, w2d, out=None):
        out.copy_(w2d)
        return out

    def gemm_h2(self, x, wp, G, M, N, K, bias=None, residual=None, act=0, out=None, out_planes=None, want_f32=True, cfg=-1, mask=None):
        self._n("gemm_h2_masked" if mask is not None else "gemm_h2")
        v = x.t @ wp.t()
        if residual is not None:
            v = v + residual.reshape(v.shape)
        v = self._mask(v, mask)
        out.copy_(v.reshape(out.shape))
        if out_planes is not None:
            self._n("planes_from_gemm")
            out_planes.t.copy_(v)
        return out, out_planes


class FakeSession(object):
    device = "cpu"

    def __init__(self):
        self.buffers, self.conv_info = {}, {}
        self.prepared = types.SimpleNamespace(enabled=False, get=lambda key, fn: fn(), join=lambda: None, weights_changed=lambda: None,
                                              refresh=lambda pre=None: pre() if pre else None)
        self.x3, self.h2, self.wino_refresh = {}, {}, (lambda: 0)

    def buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape))
        if key not in self.buffers:
            self.buffers[key] = torch.zeros(tuple(shape), dtype=dtype)
        return self.buffers[key]

    def h2_buf(self, name, rows, K):
        key = ("h2", name, rows, K)
        if key not in self.buffers:
            self.buffers[key] = Planes(rows, K)
        return self.buffers[key]

    def buf_pair(self, name, N, K):
        return self.buf("pair/" + name, (N, K))


class Net(object):
    """A small TRAIN graph in the reference's bottleneck shape (resnet_v1.py:80-113), forward values from torch, as a tape."""

    def __init__(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.g = g
        self.params, self._tape, self._requires_grad = {}, [], set()
        self.autograd_w = {}

    def conv(self, sess, x, scope, cout, k, stride=1, act=ACT_RELU, residual=None, res_stride=1, x_ref=None, res_ref=None):
        cin = x.shape[3]
        w = (torch.randn((cout, k, k, cin), generator=self.g) * (1.5 / (k * np.sqrt(cin)))).float()
        sess.conv_info[scope] = {"w": w}
        pad = (k // 2,) * 4
        wr = w.clone().double().requires_grad_(True)
        self.autograd_w[scope] = wr

        def fwd(xx, ww, rr):
            v = F.conv2d(F.pad(nchw(xx), (pad[2], pad[3], pad[0], pad[1])), w_oihw(ww), stride=stride).permute(0, 2, 3, 1)
            if rr is not None:
                v = v + rr[:, ::res_stride, ::res_stride, :][:, :v.shape[1], :v.shape[2], :]
            return torch.relu(v) if act == ACT_RELU else v
        y = fwd(x, w, residual).contiguous()
        y_ref = fwd(x_ref, wr, res_ref)
        self._tape.append(dict(kind="conv", scope=scope, x=x, y=y, k=k, stride=stride, pad=pad, act=act, residual=residual, res_stride=res_stride))
        self._requires_grad.add(y.data_ptr())
        grad_w = torch.zeros_like(w)
        self.params[scope] = types.SimpleNamespace(scope=scope, K=w[0].numel(), bias=None, grad_w=grad_w, grad_b=None, w=w, wf=w, scale=None,
                                                   acc_w=torch.zeros_like(w), acc_b=None)
        return y, y_ref


def build(sess, seed):
    net = Net(seed)
    g = torch.Generator().manual_seed(100 + seed)
    image = torch.randn((1, 8, 8, 32), generator=g)
    img_ref = image.double()
    x0, r0 = net.conv(sess, image, "conv0", 128, 3, x_ref=img_ref)                               # input of the trunk: a ReLU output
    # unit A: identity shortcut
    a1, ra1 = net.conv(sess, x0, "A/conv1", 128, 1, x_ref=r0)
    a2, ra2 = net.conv(sess, a1, "A/conv2", 128, 3, x_ref=ra1)
    xa, rxa = net.conv(sess, a2, "A/conv3", 128, 1, residual=x0, x_ref=ra2, res_ref=r0)
    # unit B: projection shortcut (no activation) + a strided 3x3; x of B has TWO consumers (conv1 and the shortcut)
    sb, rsb = net.conv(sess, xa, "B/shortcut", 256, 1, stride=1, act=ACT_NONE, x_ref=rxa)
    b1, rb1 = net.conv(sess, xa, "B/conv1", 128, 1, x_ref=rxa)
    b2, rb2 = net.conv(sess, b1, "B/conv2", 128, 3, stride=2, x_ref=rb1)
    xb, rxb = net.conv(sess, b2, "B/conv3", 256, 1, residual=sb, res_stride=2, x_ref=rb2, res_ref=rsb)
    # unit C: identity shortcut on 256 channels (the GEMM data gradients: Cin, Cout % 128 == 0)
    c1, rc1 = net.conv(sess, xb, "C/conv1", 128, 1, x_ref=rxb)
    c2, rc2 = net.conv(sess, c1, "C/conv2", 128, 3, x_ref=rc1)
    xc, rxc = net.conv(sess, c2, "C/conv3", 256, 1, residual=xb, x_ref=rc2, res_ref=rxb)
    # heads of odd width on the trunk output (two consumers of xc), like rpn_cls_score / rpn_bbox_pred
    h1, rh1 = net.conv(sess, xc, "head5", 5, 1, act=ACT_NONE, x_ref=rxc)
    h2, rh2 = net.conv(sess, xc, "head12", 12, 1, act=ACT_NONE, x_ref=rxc)
    g1 = torch.randn(h1.shape, generator=g)
    g2 = torch.randn(h2.shape, generator=g)
    loss = (rh1 * g1.double()).sum() + (rh2 * g2.double()).sum()
    loss.backward()
    return net, [(h1, g1.clone()), (h2, g2.clone())]


def run_sweep(monkeypatch, fuse, pipe, wino, h2_train, seed=0):
    from frcnn_hip import train
    ops = FakeOps()
    monkeypatch.setattr(train, "ops", ops)
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    sess = FakeSession()
    net, seeds = build(sess, seed)
    ts = train.TrainState.__new__(train.TrainState)
    ts.sess, ts.net, ts.params, ts.flat = sess, net, net.params, None
    ts._wgrad_events = None
    ts.fuse_chain, ts.pipe_dgrads = fuse, pipe
    ts.winograd = (4, 64, True) if wino else None
    ts.h2_train = h2_train
    ts.wgrad_stream, ts.wgrad_tn, ts.wgrad_h2, ts.prep_stream = 2, True, True, False
    ts._sweep(seeds, FakeStream())
    return net, ops.log


@pytest.mark.parametrize("wino", [False, True])
@pytest.mark.parametrize("h2_train", [None, 1])
def test_folded_passes_give_the_bits_of_the_separate_passes_and_the_gradients_of_autograd(monkeypatch, wino, h2_train):
    n0, log0 = run_sweep(monkeypatch, False, False, wino, h2_train)
    n1, log1 = run_sweep(monkeypatch, True, False, wino, h2_train)
    n2, log2 = run_sweep(monkeypatch, True, True, wino, h2_train)
    assert set(n0.params) == set(n1.params) and len(n0.params) == 13
    for sc in n0.params:
        ref = n0.autograd_w[sc].grad.float()
        scale = float(ref.abs().max())
        assert scale > 0
        assert float((n0.params[sc].grad_w - ref).abs().max()) <= 2e-5 * scale, sc        # the sweep itself (separate passes)
        assert torch.equal(n0.params[sc].grad_w, n1.params[sc].grad_w), sc                # folding changes no bit
        assert float((n2.params[sc].grad_w - ref).abs().max()) <= 2e-5 * scale, sc        # other data-gradient forms: rounding only
    # what the folded sweep no longer launches
    assert log0["relu_bwd"] == 10                                                        # one per ReLU record
    assert log1.get("relu_bwd", 0) <= 3 and log1.get("conv2d_masked", 0) + log1.get("gemm_h2_masked", 0) + log1.get("wino_masked", 0) >= 7
    if h2_train:
        assert log1.get("h2_split", 0) < log0["h2_split"] and log1.get("planes_from_gemm", 0) + log1.get("planes_from_wino", 0) >= 1
    assert log2.get("dgrad_strided", 0) < log1["dgrad_strided"]                           # strided 3x3 + odd-width heads left the gather form
    assert log0["wgrad"] == log1["wgrad"] == log2["wgrad"] == 13


def test_a_tensor_with_an_unmasked_contribution_keeps_its_relu_pass(monkeypatch):
    """B's input feeds conv1 (masked data gradient) AND the projection shortcut; whichever lands second accumulates -- the sum must end up
    masked exactly once more or already be masked; the heads' two data gradients into the trunk output likewise.  Checked through the
    gradients of the layers BELOW those tensors (A/*, conv0) in the test above; here: the call pattern."""
    n1, log1 = run_sweep(monkeypatch, True, True, True, 1, seed=3)
    n0, log0 = run_sweep(monkeypatch, False, True, True, 1, seed=3)
    for sc in n0.params:
        assert torch.equal(n0.params[sc].grad_w, n1.params[sc].grad_w), sc
    assert log1.get("relu_bwd", 0) <= 1 < log0["relu_bwd"]          # every contribution came masked out of its launch


def run_steps(monkeypatch, in_sweep, steps=3, chunk=3):
    """`steps` x (reverse sweep + solver) on one tape: the in-sweep solver updates ranges of the descriptor table while the sweep goes on"""
    from frcnn_hip import train
    ops = FakeOps()
    monkeypatch.setattr(train, "ops", ops)
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    sess = FakeSession()
    net, seeds = build(sess, 5)
    ts = train.TrainState.__new__(train.TrainState)
    ts.sess, ts.net, ts.params, ts.flat = sess, net, net.params, None
    ts._wgrad_events, ts.reg_scopes = None, []
    ts.momentum, ts.weight_decay, ts.double_bias, ts.bias_decay = 0.9, 1e-4, False, False
    ts.fuse_chain, ts.pipe_dgrads, ts.winograd, ts.h2_train = True, True, (4, 64, True), 1
    ts.wgrad_stream, ts.wgrad_tn, ts.wgrad_h2, ts.prep_stream = 2, True, True, False
    ts.solver_in_sweep, ts.SOLVER_CHUNK, ts.lr = in_sweep, chunk, 1e-4
    order = []
    for _ in range(steps):
        ops.updated = []
        ts._sweep([(t, g.clone()) for t, g in seeds], FakeStream(), in_sweep)
        ts.apply(ts.lr)
        order.append(list(ops.updated))
    return net, ops.log, order


def test_the_solver_inside_the_sweep_updates_every_tensor_exactly_once_with_the_same_result(monkeypatch):
    """TrainState: from the second step on the solver updates ranges of its descriptor table on a side stream while the sweep goes on
    (frcnn_sgd_momentum_range), always behind the data gradient that last reads those filters; apply() updates the rest.  Three steps with
    and without it on the same tape: every master filter and momentum slot bit-identical, every tensor updated exactly once per step, the
    first step (no descriptor table yet) in one launch."""
    n0, log0, order0 = run_steps(monkeypatch, False)
    n1, log1, order1 = run_steps(monkeypatch, True)
    for sc in n0.params:
        assert torch.equal(n0.params[sc].w, n1.params[sc].w), sc
        assert torch.equal(n0.params[sc].acc_w, n1.params[sc].acc_w), sc
        assert float(n0.params[sc].acc_w.abs().max()) > 0
    assert log0["sgd_range"] == 3                                        # one launch per step
    assert log1["sgd_range"] == 1 + 2 * 5                                # step 1: one; then 13 tensors in chunks of 3 + the rest in apply()
    for net, order in ((n0, order0), (n1, order1)):
        every = sorted(p.w.data_ptr() for p in net.params.values())
        for upd in order:
            assert sorted(upd) == every                                  # each tensor once per step, none skipped, none twice
    # in-sweep order: the END of the table (the layers the sweep visits first) is updated first
    table_order = [p.w.data_ptr() for p in n1.params.values()]
    assert order1[1][:3] == table_order[-3:] or set(order1[1][:3]) <= set(table_order[-4:])


class FakeExchange(object):
    """stand-in for parallel.BucketedAllReduce: the same range bookkeeping, the 'all-reduce' doubles the range in place (two identical
    replicas: sum = 2 g) and logs what it was given; wait_sent logs who waited"""
    multi_stream = True

    def __init__(self, bucket):
        self.bucket, self.sent_from, self.final_from, self.log = bucket, None, None, []

    def ready(self, flat, data_ptr, stream=None, streams=None):
        n = flat.numel()
        if self.sent_from is None:
            self.sent_from = n
        off = (int(data_ptr) - flat.data_ptr()) // 4
        assert self.final_from is None or off <= self.final_from
        self.final_from = off
        while self.sent_from - self.final_from >= self.bucket:
            lo = self.sent_from - self.bucket
            flat[lo:self.sent_from] *= 2.0
            self.log.append(("send", lo, self.sent_from, len(streams or [])))
            self.sent_from = lo

    def wait_sent(self, stream):
        self.log.append(("wait", self.sent_from))

    def finish(self, flat):
        hi = flat.numel() if self.sent_from is None else self.sent_from
        if hi > 0:
            flat[0:hi] *= 2.0
            self.log.append(("send", 0, hi, 0))
        self.sent_from = self.final_from = None
        return flat


def run_dp_steps(monkeypatch, in_sweep, steps=3, chunk=3, bucket=40000):
    from frcnn_hip import train
    ops = FakeOps()
    monkeypatch.setattr(train, "ops", ops)
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    sess = FakeSession()
    net, seeds = build(sess, 5)
    total = sum(p.grad_w.numel() for p in net.params.values())
    flat, off = torch.zeros(total), 0
    for p in net.params.values():                            # the flat gradient buffer in forward order (TrainState.build)
        n = p.grad_w.numel()
        p.grad_w = flat[off:off + n].view(p.grad_w.shape)
        off += n
    ts = train.TrainState.__new__(train.TrainState)
    ts.sess, ts.net, ts.params, ts.flat = sess, net, net.params, flat
    ts._wgrad_events, ts.reg_scopes = None, []
    ts.momentum, ts.weight_decay, ts.double_bias, ts.bias_decay = 0.9, 1e-4, False, False
    ts.fuse_chain, ts.pipe_dgrads, ts.winograd, ts.h2_train = True, True, (4, 64, True), 1
    ts.wgrad_stream, ts.wgrad_tn, ts.wgrad_h2, ts.prep_stream = 2, True, True, False
    ts.solver_in_sweep, ts.SOLVER_CHUNK, ts.lr = in_sweep, chunk, 1e-4
    ts.world_size, ts.all_reduce = 2, FakeExchange(bucket)
    order, sends = [], []
    for _ in range(steps):
        ops.updated, ts.all_reduce.log = [], []
        ts._sweep([(t, g.clone()) for t, g in seeds], FakeStream(), in_sweep)
        ts.apply(ts.lr, world_size=2, all_reduce=ts.all_reduce)
        order.append(list(ops.updated))
        sends.append(list(ts.all_reduce.log))
    return net, ops.log, order, sends, total


def test_data_parallel_sweep_updates_behind_the_exchange_and_equals_the_update_after_the_sweep(monkeypatch):
    """Data parallel (world 2) with the bucketed exchange: the in-sweep solver may only touch parameters whose gradients the exchange has
    already taken ([sent_from, end) of the flat buffer), waits for those collectives first, folds the mean (1 / world) into the update,
    and apply() finishes the rest -- the result equals the solver after the sweep bit for bit, every tensor once per step, and the
    exchange is told about BOTH filter-gradient streams."""
    n0, log0, order0, sends0, total = run_dp_steps(monkeypatch, False)
    n1, log1, order1, sends1, _ = run_dp_steps(monkeypatch, True)
    for sc in n0.params:
        assert torch.equal(n0.params[sc].w, n1.params[sc].w), sc
        assert torch.equal(n0.params[sc].acc_w, n1.params[sc].acc_w), sc
    every = sorted(p.w.data_ptr() for p in n1.params.values())
    for upd in order1:
        assert sorted(upd) == every
    assert log0["sgd_range"] == 3 and log1["sgd_range"] > 3 + 2          # in-sweep launches from step 2 on
    step = sends1[1]
    ranges = [e for e in step if e[0] == "send"]
    assert ranges[0][2] == total and ranges[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(ranges, ranges[1:]))      # the buffer, back to front, no gap
    assert any(e[3] == 2 for e in ranges)                                   # ordered after both filter-gradient streams
    waits = [e for e in step if e[0] == "wait"]
    assert waits, "the in-sweep solver never waited for the exchange"
    # every in-sweep update came after a wait, and covered only offsets >= what had been sent by then
    offs = {p.w.data_ptr(): (p.grad_w.data_ptr() - n1.params[next(iter(n1.params))].grad_w.data_ptr()) // 4 for p in n1.params.values()}
    sent_at_wait = min(w[1] for w in waits)
    in_sweep_updates = order1[1][:len(order1[1]) - 1]
    assert all(offs[w] >= sent_at_wait for w in in_sweep_updates[:3])
