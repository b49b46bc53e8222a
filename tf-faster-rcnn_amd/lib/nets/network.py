"""Network -- the Faster R-CNN graph of the reference (lib/nets/network.py), re-hosted on
libfrcnn_hip.so.  No TensorFlow: `sess` is a frcnn_hip.runtime.Session (device + stream + weights),
`create_architecture` declares the variables (TF/slim names and layouts), and one call of
`test_image` runs image -> backbone -> RPN -> proposals -> RoI crop -> tail -> cls/bbox entirely
on the GPU -- the py_func host hops of network.py:100-126 and the per-image sess.run (:470-479)
are gone; the chain is captured once per image shape into a hipGraph and replayed.

Public surface kept from the reference (file:line = /root/reference/lib/nets/network.py):
  create_architecture (386), test_image (470), extract_head (464), hooks _image_to_head /
  _head_to_tail (380-384), _anchor_component (210), _region_proposal (323), _crop_pool_layer (141),
  _region_classification (361), self._predictions keys (348-353, 373-376).
"""
import collections

import numpy as np
import torch

from frcnn_hip import ACT_NONE, ACT_RELU, NMS_RULE_CPU, NMS_RULE_GPU, ops
from frcnn_hip.runtime import VarSpec
from model.config import cfg


class Network(object):
    def __init__(self):
        self._predictions = {}
        self._losses = {}
        self._anchor_targets = {}
        self._proposal_targets = {}
        self._layers = {}
        self._gt_image = None
        self._act_summaries = []
        self._score_summaries = {}
        self._train_summaries = []
        self._event_summaries = {}
        self._variables_to_fix = {}
        self._feat_stride = [16, ]
        self._scope = "network"
        self._sess = None
        self._image = None
        self._im_info = None
        self._var_specs = collections.OrderedDict()
        self._tape = []
        self._requires_grad = set()
        self._gt_boxes = None
        self._sample_seed = 0
        self.replay_stats = dict(eager=0, recorded=0, replayed=0)      # train_step_async under cfg.HIP.TRAIN_REPLAY
        self._train_state = None
        self._fuse_tail_entry = False          # TEST-only graph restructuring, see resnetv1._fused_tail_entry
        self._h2_of = {}                       # activation address -> ops.H2 operand planes of that tensor (cfg.HIP.MFMA_H2)
        self._f32_missing = set()              # addresses of activations that exist ONLY as operand planes (never read as f32)

    # ------------------------------------------------------------------ variable declaration
    def _var(self, name, shape, init, arg=None):
        self._var_specs[name] = VarSpec(shape, init, arg)

    def _declare_conv_bn(self, scope, kh, kw, cin, cout, residual_branch_end=False):
        self._var(scope + "/weights", (kh, kw, cin, cout), "he")
        self._var(scope + "/BatchNorm/gamma", (cout,), "bn_gamma_res" if residual_branch_end else "bn_gamma")
        self._var(scope + "/BatchNorm/beta", (cout,), "bn_beta")
        self._var(scope + "/BatchNorm/moving_mean", (cout,), "bn_mean")
        self._var(scope + "/BatchNorm/moving_variance", (cout,), "bn_var")

    def _declare_conv_bias(self, scope, kh, kw, cin, cout, std):
        self._var(scope + "/weights", (kh, kw, cin, cout), "normal", std)
        self._var(scope + "/biases", (cout,), "zeros")

    def _declare_backbone(self):
        raise NotImplementedError

    def _head_channels(self):
        raise NotImplementedError

    def _tail_channels(self):
        raise NotImplementedError

    def variable_specs(self):
        return self._var_specs

    # ------------------------------------------------------------------ building blocks
    def _conv(self, x, scope, k, stride=1, pad=(0, 0, 0, 0), act=ACT_RELU, bn_eps=None, residual=None,
              res_stride=1, fold_w=False, out_affine=None, real_cin=None, no_bias=False, emit_h2=False, want_f32=True):
        """emit_h2 / want_f32 (cfg.HIP.MFMA_H2, TEST mode): the caller knows the consumers of the result -- emit_h2: a frcnn_gemm_h2
        launch will read it, so the producer writes its operand planes from its epilogue; want_f32 = False: nothing reads the
        float32 tensor (honoured only where the planes are emitted)."""
        sess = self._sess
        N, H, W, Cin = x.shape
        wino = (cfg.HIP.WINOGRAD and k == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1)
                and residual is None and not fold_w and out_affine is None and not no_bias
                and Cin % 32 == 0 and Cin >= cfg.HIP.WINOGRAD_MIN_CIN and act in (ACT_NONE, ACT_RELU)
                and not any(tok in scope for tok in cfg.HIP.WINOGRAD_DIRECT_SCOPES))
        if wino and self._mode == "TEST":
            return self._conv_winograd(x, scope, act, bn_eps, emit_h2, want_f32)
        wino = wino and cfg.HIP.WINOGRAD_TRAIN
        w, b = sess.conv_params(scope, bn_eps=bn_eps, fold_w=fold_w,
                                out_scale=None if out_affine is None else out_affine[0],
                                out_shift=None if out_affine is None else out_affine[1])
        if no_bias:
            b = None
        N, H, W, Cin = x.shape
        OH = ops.conv_out_size(H, k, stride, pad[0], pad[1])
        OW = ops.conv_out_size(W, k, stride, pad[2], pad[3])
        Cout = w.shape[0]
        out = sess.buf(self._tag + "/" + scope, (N, OH, OW, Cout))
        flops = 2 * N * OH * OW * Cout * k * k * (Cin if real_cin is None else real_cin)
        plain = k == 1 and stride == 1 and tuple(pad) == (0, 0, 0, 0) and not fold_w and (residual is None or res_stride == 1)
        M = N * OH * OW
        if wino:
            # TRAIN: the filter changes every step -> transform the live (folded) device filter, then the same Winograd chain
            m = self._winograd_scheme(scope, H, W)
            G, T = ops.winograd_points(m), ops.winograd_tiles(N, H, W, m)
            # (weight-only: after the first step the solver has it re-run beside the forward pass, frcnn_hip/runtime.py PreparedFilters)
            u = sess.prepared.get(("fwd", "wino_u", self._tag, scope, m), lambda: ops.winograd_filter_transform_device(
                w, m, False, out=sess.buf(self._tag + "/wino_u/" + scope, (G, Cout, Cin))))
            mm = sess.buf(self._tag + "/wino_m", (G, T, Cout))
            v = sess.buf(self._tag + "/wino_v", (G, T, Cin))
            # emit_h2: the output transform writes the float32 tensor (the tape needs it) AND the operand planes of the 1x1 that follows
            yp = sess.h2_buf(self._tag + "/" + scope, M, Cout) if (emit_h2 and cfg.HIP.MFMA_H2 and Cout % 128 == 0) else None
            sess.mark("conv:" + scope, 2 * G * T * Cout * Cin,
                      lambda: ops.conv3x3_winograd(x, u, b, act, out=out, v_buf=v, m_buf=mm, out_planes=yp),
                      nbytes=4 * (v.numel() + u.numel() + mm.numel()))
            self._wrote(out, yp, True) if yp is not None else self._wrote(out)
        elif plain and self._h2_eligible(M, Cout, Cin, 1, scope) and self._h2_input(x) is not None:
            # a plain GEMM with a static filter on the fp16 matrix pipe, block-scaled two-piece operands (cfg.HIP.MFMA_H2)
            xp, wp = self._h2_input(x), sess.h2_planes(w)
            yp = sess.h2_buf(self._tag + "/" + scope, M, Cout) if emit_h2 else None
            y = out if (want_f32 or yp is None or self._mode == "TRAIN") else None
            res = residual
            if residual is not None and residual.data_ptr() in self._f32_missing:
                res = self._h2_of[residual.data_ptr()]              # the trunk exists as operand planes only (cfg.HIP.H2_TRUNK_PLANES)
            tcfg = int(cfg.HIP.H2_TILE_CFG)
            sess.mark("conv:h2:" + scope, flops, lambda: ops.gemm_h2(xp, wp, 1, M, Cout, Cin, b, res, act, out=y, out_planes=yp,
                                                                      want_f32=False, cfg=tcfg),
                      nbytes=4 * M * Cin + 4 * w.numel() + (4 * out.numel() if y is not None else 0)
                      + (4 * out.numel() if yp is not None else 0) + (4 * out.numel() if residual is not None else 0))
            self._wrote(out, yp, y is not None)
        elif plain and self._x3_eligible(M, Cout, Cin, 1):
            # ... on the bf16 matrix pipe with exact bf16x3 operand splits (cfg.HIP.MFMA_X3)
            self._need_f32(x), self._need_f32(residual)
            planes = sess.x3_planes(w)
            xc = int(cfg.HIP.X3_TILE_CFG)
            sess.mark("conv:x3:" + scope, flops, lambda: ops.gemm_x3(x, planes, 1, M, Cout, Cin, b, residual, act, out=out, cfg=xc),
                      nbytes=4 * (x.numel() + out.numel() + (out.numel() if residual is not None else 0)) + 6 * w.numel())
            self._wrote(out)
        else:
            self._need_f32(x), self._need_f32(residual)
            sess.mark("conv:" + scope, flops,
                      lambda: ops.conv2d(x, w, b, k, k, stride, pad, act, residual, res_stride, fold_w, out=out),
                      nbytes=4 * (x.numel() + w.numel() + out.numel() + (out.numel() if residual is not None else 0)))
            self._wrote(out)
        if self._mode == "TRAIN":
            self._tape.append(dict(kind="conv", scope=scope, x=x, y=out, k=k, stride=stride, pad=tuple(pad), act=act,
                                   residual=residual, res_stride=res_stride))
            if self.trainable_scope(scope) or x.data_ptr() in self._requires_grad:
                self._requires_grad.add(out.data_ptr())
            if residual is not None and residual.data_ptr() in self._requires_grad:
                self._requires_grad.add(out.data_ptr())
        return out

    # ---- cfg.HIP.MFMA_H2 plumbing -------------------------------------------------------------------------------------------------
    H2_MIN_CHANNEL_RATIO = 2.0 ** -18      # see Session.h2_channel_spread

    def _h2_eligible(self, M, N, K, G, scope=None):
        """K % 128 == 0 (scale blocks), N % 128 == 0 (tiles), enough tiles to fill the chip, and the 32-bit offset limits of
        frcnn_gemm_h2.  TEST mode: static filters, split once.  TRAIN mode (cfg.HIP.H2_TRAIN): the pointwise convolutions of the
        forward pass take the same kernel -- the solver re-splits the updated filters after every step (Session.h2_refresh), the
        float32 outputs the tape needs are always written, inputs without planes are split by a separate pass."""
        rows = G * M
        if scope is not None and self._mode == "TEST" and self._sess.h2_channel_spread(scope) < self.H2_MIN_CHANNEL_RATIO:
            return False              # a filter whose entries for one input channel are < 2^-18 of the rest: exact x3 split instead
        return (bool(cfg.HIP.MFMA_H2) and (self._mode == "TEST" or bool(cfg.HIP.H2_TRAIN)) and K % 128 == 0 and N % 128 == 0
                and ((self._plan_rows(M) + 127) // 128) * (N // 128) * G >= self._h2_min_tiles()
                and 4 * rows * K < (1 << 32) and 4 * N * K < (1 << 32) and M * N < (1 << 29))

    @staticmethod
    def h2_min_tiles(mode):
        """Tiles a launch must have to take frcnn_gemm_h2.  TEST: cfg.HIP.H2_MIN_TILES.  TRAIN (one image per step): cfg.HIP.H2_TRAIN_MIN_TILES,
        or -- when that is None -- the same knob as TEST mode (tests force the kernel onto toy TRAIN networks with H2_MIN_TILES = 1 and
        H2_TRAIN_MIN_TILES = None).  Each value means what it says: no comparison against a default."""
        if mode == "TRAIN" and cfg.HIP.H2_TRAIN_MIN_TILES is not None:
            return int(cfg.HIP.H2_TRAIN_MIN_TILES)
        return int(cfg.HIP.H2_MIN_TILES)

    def _h2_min_tiles(self):
        return self.h2_min_tiles(self._mode)

    def _h2_input(self, x):
        """Operand planes of activation x: those its producer emitted, else (cfg.HIP.H2_LAZY_SPLIT) a frcnn_h2_split pass, else None."""
        xp = self._h2_of.get(x.data_ptr())
        if xp is None and cfg.HIP.H2_LAZY_SPLIT:
            self._need_f32(x)
            K = x.shape[-1]
            xp = self._sess.h2_buf(self._tag + "/split@%x" % x.data_ptr(), x.numel() // K, K)
            self._sess.mark("op:h2_split", 0, lambda: ops.h2_split(x, out=xp), nbytes=8 * x.numel())
            self._h2_of[x.data_ptr()] = xp
        return xp

    def _wrote(self, t, planes=None, f32=True):
        """Every launch that writes activation t reports it: planes derived from an earlier write are dropped (buffers are static
        and sub-graphs may be re-run on new inputs), planes this launch emitted are registered."""
        k = t.data_ptr()
        if planes is None:
            self._h2_of.pop(k, None)
        else:
            self._h2_of[k] = planes
        (self._f32_missing.discard if f32 else self._f32_missing.add)(k)
        return t

    def _need_f32(self, x):
        if x is not None and x.data_ptr() in self._f32_missing:
            raise RuntimeError("graph construction error: a float32 consumer reads a tensor that was emitted as operand planes only")

    def _x3_eligible(self, M, N, K, G):
        """cfg.HIP.MFMA_X3: TEST mode (static filters), N % 64 == 0, K % 32 == 0 and at least 150 tiles of 128 x 128 -- below that
        the split-K f32 launches are as fast (profiles/r02_m_x3_sweep.txt, single-image rows)."""
        return (bool(cfg.HIP.MFMA_X3) and self._mode == "TEST" and N % 64 == 0 and K % 32 == 0
                and ((self._plan_rows(M) + 127) // 128) * ((N + 127) // 128) * G >= 150 and M * N < (1 << 31) and N * K < (1 << 28))

    @staticmethod
    def _winograd_scheme(scope, H, W):
        """2 / 4 = F(m x m,3x3) tiles; 7 = the mixed F(4,3)+F(3,3) scheme for 7x7 maps (121 instead of 144 GEMM rows per RoI)."""
        if any(tok in scope for tok in cfg.HIP.WINOGRAD_F2_SCOPES):
            return 2
        m = int(cfg.HIP.WINOGRAD_M)
        return 7 if (m == 4 and H == 7 and W == 7 and cfg.HIP.WINOGRAD_7X7) else m

    def _conv_winograd(self, x, scope, act, bn_eps, emit_h2=False, want_f32=True):
        """3x3 / stride 1 / SAME convolution as Winograd F(m x m,3x3): input transform -> (m+2)^2 GEMMs in ONE launch of
        the f32-MFMA kernel -> output transform with bias + ReLU.  Exact algebra in f32; m = 2 (2.25x fewer
        multiplications, rounding like the direct kernel) or m = 4 (4x fewer; a single layer rounds ~10x worse than
        direct, but through the full ResNet-101 the outputs move by ~1e-6 relative, profiles/r01_e_winograd_error.txt).
        cfg.HIP.MFMA_H2: the input transform emits V as operand planes, the products run in frcnn_gemm_h2, and (emit_h2) the
        output transform emits the result as the next 1x1 convolution's operand planes."""
        sess = self._sess
        self._need_f32(x)
        N, H, W, Cin = x.shape
        m = self._winograd_scheme(scope, H, W)
        u, b = sess.winograd_params(scope, bn_eps=bn_eps, m=m)
        G, Cout = u.shape[0], u.shape[1]
        T = ops.winograd_tiles(N, H, W, m)
        mm = sess.buf(self._tag + "/wino_m", (G, T, Cout))
        out = sess.buf(self._tag + "/" + scope, (N, H, W, Cout))
        flops = 2 * G * T * Cout * Cin
        if self._h2_eligible(T, Cout, Cin, G, scope):
            vp, wp = sess.h2_buf(self._tag + "/wino_v", G * T, Cin), sess.h2_planes(u)
            sess.mark("op:wino_in", 0, lambda: ops.winograd_input_transform_h2(x, vp, m), nbytes=4 * (x.numel() + G * T * Cin))
            tcfg = int(cfg.HIP.H2_TILE_CFG)
            sess.mark("conv:h2:" + scope, flops, lambda: ops.gemm_h2(vp, wp, G, T, Cout, Cin, out=mm, cfg=tcfg),
                      nbytes=4 * (G * T * Cin + mm.numel()) + 4 * u.numel())
        else:
            v = sess.buf(self._tag + "/wino_v", (G, T, Cin))
            sess.mark("op:wino_in", 0, lambda: ops.winograd_input_transform(x, v, m), nbytes=4 * (x.numel() + v.numel()))
            if self._x3_eligible(T, Cout, Cin, G):
                planes = sess.x3_planes(u)
                xc = int(cfg.HIP.X3_TILE_CFG)
                sess.mark("conv:x3:" + scope, flops, lambda: ops.gemm_x3(v, planes, G, T, Cout, Cin, out=mm, cfg=xc),
                          nbytes=4 * (v.numel() + mm.numel()) + 6 * u.numel())
            else:
                sess.mark("conv:" + scope, flops, lambda: ops.gemm_batched_nt(v, u, mm), nbytes=4 * (v.numel() + u.numel() + mm.numel()))
        if emit_h2 and cfg.HIP.MFMA_H2 and Cout % 128 == 0:
            yp = sess.h2_buf(self._tag + "/" + scope, N * H * W, Cout)
            y = out if want_f32 else None
            sess.mark("op:wino_out", 0, lambda: ops.winograd_output_transform_h2(mm, b, act, (N, H, W, Cout), m, yp, y),
                      nbytes=4 * (mm.numel() + out.numel() * (2 if want_f32 else 1)))
            self._wrote(out, yp, y is not None)
        else:
            sess.mark("op:wino_out", 0, lambda: ops.winograd_output_transform(mm, b, act, out, m), nbytes=4 * (mm.numel() + out.numel()))
            self._wrote(out)
        return out

    def _conv1x1_mean(self, x, scope, group_rows, act=ACT_RELU, bn_eps=None, residual=None, name="fc7"):
        """The tail's last 1x1 convolution fused with reduce_mean over the P*P positions of every RoI (resnet_v1.py:115-125), TEST mode,
        cfg.HIP.FUSE_TAIL_MEAN: frcnn_gemm_h2_mean -- the [R*P*P, Cout] tensor is never written or re-read.  One batch entry per IMAGE:
        a RoI's rows are added in an order that depends on its index inside its image only, so batch slots and batch sizes do not
        change a bit of fc7 (the caller checks _mean_fusable first: the layer must be h2-eligible and x must exist as operand planes)."""
        sess = self._sess
        w, b = sess.conv_params(scope, bn_eps=bn_eps)
        Cin, Cout = x.shape[-1], w.shape[0]
        rows = x.numel() // Cin
        G = max(1, getattr(self, "_plan_batch", 0))
        assert rows % G == 0 and (rows // G) % group_rows == 0
        M = rows // G
        out = sess.buf(self._tag + "/" + name, (rows // group_rows, Cout))
        xp, wp = self._h2_of[x.data_ptr()], sess.h2_planes(w)
        res = residual
        if residual is not None and residual.data_ptr() in self._f32_missing:
            res = self._h2_of[residual.data_ptr()]
        tcfg = int(cfg.HIP.H2_TILE_CFG)
        sess.mark("conv:h2:" + scope, 2 * rows * Cout * Cin, lambda: ops.gemm_h2_mean(xp, wp, G, M, Cout, Cin, b, res, act, group_rows, out=out, cfg=tcfg),
                  nbytes=4 * (rows * Cin + w.numel() + out.numel() + (rows * Cout if residual is not None else 0)))
        return self._wrote(out)

    def _mean_fusable(self, rows, cout, cin, scope, group_rows=1):
        """cfg.HIP.FUSE_TAIL_MEAN applies where the tail's last convolution runs in frcnn_gemm_h2 (TEST mode, h2-eligible shape and filter)
        and the RoI rows split evenly over the images of the batch (whole groups of `group_rows` rows per image)."""
        G = max(1, getattr(self, "_plan_batch", 0))
        return (bool(cfg.HIP.FUSE_TAIL_MEAN) and self._mode == "TEST" and rows % G == 0 and (rows // G) % max(1, int(group_rows)) == 0
                and self._h2_eligible(rows, cout, cin, 1, scope)
                and bool(cfg.HIP.H2_LAZY_SPLIT or cfg.HIP.WINOGRAD))         # its input must exist as planes: emitted by the Winograd conv2, or split lazily

    # ------------------------------------------------------------------ ImageNet-pretrained weights (train_val.py:177-202)
    _rgb_first_conv = None                   # scope tail of the stem conv whose input channels are RGB in the released weights

    def get_variables_to_restore(self, variables, var_keep_dic):
        """Names to restore verbatim from a pretrained checkpoint: everything the checkpoint has, minus the variables
        fix_variables() rewrites (resnet_v1.py:154-167, vgg16.py:62-79, mobilenet_v1.py:253-264)."""
        skip = set(self._variables_to_fix_names())
        self._variables_to_fix = {}
        keep = []
        for name in variables:
            if name in skip:
                self._variables_to_fix[name] = True
                continue
            if name in var_keep_dic:
                keep.append(name)
        return keep

    def _variables_to_fix_names(self):
        return [] if self._rgb_first_conv is None else [self._scope + self._rgb_first_conv + "/weights"]

    def fix_variables(self, sess, pretrained_model):
        """RGB -> BGR on the stem filter (`tf.reverse(conv1_rgb, [2])`, resnet_v1.py:168-178); subclasses add their own."""
        from frcnn_hip.tensor_bundle import open_checkpoint
        reader = open_checkpoint(pretrained_model)
        fixed = {}
        for name in self._variables_to_fix_names():
            fixed[name] = self._fix_one(name, reader.get_tensor(name))
        sess.load_variables(fixed)
        sess.conv_info.clear()
        return sorted(fixed)

    def _fix_one(self, name, value):
        return np.ascontiguousarray(value[:, :, ::-1, :])

    def trainable_scope(self, scope):
        """Which filters the solver updates (reference: `trainable=` flags of the slim layers)."""
        return True

    def _reshape_layer(self, bottom, num_dim, name):
        raise NotImplementedError("folded into frcnn_rpn_softmax (network.py:68-78 only served the pair softmax)")

    def _softmax_layer(self, bottom, name):
        if name.startswith("rpn_cls_prob"):
            A = self._num_anchors
            out = self._sess.buf(self._tag + "/" + name, bottom.shape[:3] + (2 * A,))
            return self._sess.mark("op:rpn_softmax", 0, lambda: ops.rpn_softmax(bottom, A, out=out), nbytes=8 * out.numel())
        out = self._sess.buf(self._tag + "/" + name, bottom.shape)
        return self._sess.mark("op:softmax_rows", 0, lambda: ops.softmax_rows(bottom, out=out), nbytes=8 * out.numel())

    # ------------------------------------------------------------------ graph pieces (same names as the reference)
    def _anchor_component(self):
        # network.py:210-231: feature-map size from im_info; anchors are regenerated in-kernel by the
        # proposal layer, so only the float64 base anchors [A,4] live on the device.
        h = int(np.ceil(self._im_info[0] / np.float32(self._feat_stride[0])))
        w = int(np.ceil(self._im_info[1] / np.float32(self._feat_stride[0])))
        base = ops.generate_anchors(16, self._anchor_ratios, self._anchor_scales)
        if cfg.USE_E2E_TF:
            # generate_anchors_pre_tf (snippets.py:44): tf.constant(anchors, dtype=tf.int32) truncates the base anchors
            base = np.trunc(base)
        key = ("base_anchors", tuple(self._anchor_ratios), tuple(self._anchor_scales), bool(cfg.USE_E2E_TF))
        if key not in self._sess.packed:
            self._sess.packed[key] = self._sess.to_device(base, torch.float64)
        self._base_anchors = self._sess.packed[key]
        self._anchor_length = h * w * self._num_anchors
        self._anchors = None      # materialise on demand with ops.generate_anchors_pre

    @staticmethod
    def _nms_rule():
        """lib/model/nms_wrapper.py:15-23: cfg.USE_GPU_NMS picks the reference's CUDA kernel (`ovr > thresh`,
        nms_kernel.cu:71), otherwise the Cython cpu_nms rule (`ovr >= thresh`, cpu_nms.pyx:65) -- same HIP kernels."""
        return NMS_RULE_GPU if cfg.USE_GPU_NMS else NMS_RULE_CPU

    def _proposal_layer(self, rpn_cls_prob, rpn_bbox_pred, name):
        """network.py:110-131 -> frcnn_proposal_layer_batched: ONE set of launches for the B images of the batch (the
        reference graph is batch-1; a batch here is B independent images whose launches are shared).  rois[:,0] = image index."""
        c = cfg[self._mode]
        post = int(c.RPN_POST_NMS_TOP_N)
        s = self._sess
        B = rpn_cls_prob.shape[0]
        rois = s.buf(self._tag + "/rois", (B * post, 5))
        scores = s.buf(self._tag + "/roi_scores", (B * post, 1))
        num = s.buf(self._tag + "/num_rois", (B,), torch.int32)
        if cfg.USE_E2E_TF:
            # network.py:112-121 -> proposal_layer_tf: tf.image.non_max_suppression over ALL anchors, no pre-NMS top-N
            s.mark("op:proposal_layer_tf", 0, lambda: ops.proposal_layer_tf(
                rpn_cls_prob, rpn_bbox_pred, self._im_info[0], self._im_info[1], self._feat_stride[0],
                self._base_anchors, post, float(c.RPN_NMS_THRESH), rois=rois, scores=scores, num=num))
        else:
            N = rpn_cls_prob.shape[1] * rpn_cls_prob.shape[2] * self._num_anchors
            s.mark("op:proposal_layer", 0, lambda: ops.proposal_layer(
                rpn_cls_prob, rpn_bbox_pred, self._im_info[0], self._im_info[1], self._feat_stride[0],
                self._base_anchors, int(c.RPN_PRE_NMS_TOP_N), post, float(c.RPN_NMS_THRESH), rois=rois, scores=scores, num=num,
                rule=self._nms_rule()), nbytes=B * 36 * N)
        self._num_rois = num
        self._rois_per_image = post
        return rois, scores

    def _proposal_top_layer(self, rpn_cls_prob, rpn_bbox_pred, name):
        # network.py:88-108.  USE_E2E_TF's proposal_top_layer_tf (tf.nn.top_k: descending, equal scores -> lower index first)
        # selects and orders exactly like the kernel's (score desc, index asc) keys, so both settings share it.
        n, s = int(cfg.TEST.RPN_TOP_N), self._sess
        assert rpn_cls_prob.shape[0] == 1, "TEST.MODE 'top' is provided for single images"
        self._rois_per_image = n
        rois, scores = ops.proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, self._im_info[0], self._im_info[1],
                                              self._feat_stride[0], self._base_anchors, n,
                                              rois=s.buf(self._tag + "/top_rois", (n, 5)),
                                              scores=s.buf(self._tag + "/top_scores", (n, 1)))
        self._num_rois = None
        return rois, scores

    def _anchor_target_layer(self, rpn_cls_score, name):
        # network.py:162-183 -> lib/layer_utils/anchor_target_layer.py, on device
        _, H, W, _ = rpn_cls_score.shape
        t = cfg.TRAIN
        labels, tg, iw, ow = ops.anchor_target_layer(self._gt_boxes, self._im_info[0], self._im_info[1], H, W, self._base_anchors,
                                                     self._feat_stride[0], t.RPN_BATCHSIZE, t.RPN_FG_FRACTION, t.RPN_POSITIVE_OVERLAP,
                                                     t.RPN_NEGATIVE_OVERLAP, seed=self._sample_seed,
                                                     opts=ops.rpn_target_opts(t.RPN_CLOBBER_POSITIVES, t.RPN_POSITIVE_WEIGHT,
                                                                              t.RPN_BBOX_INSIDE_WEIGHTS))
        self._anchor_targets = dict(rpn_labels=labels, rpn_bbox_targets=tg, rpn_bbox_inside_weights=iw, rpn_bbox_outside_weights=ow)
        return labels

    def _proposal_target_layer(self, rois, roi_scores, name):
        # network.py:185-208 -> lib/layer_utils/proposal_target_layer.py, on device
        t = cfg.TRAIN
        # the padded proposal buffer goes in whole; the kernel reads the valid row count on the device (no host sync here)
        out = ops.proposal_target_layer(rois, roi_scores.view(-1), self._gt_boxes, self._num_classes,
                                        t.BATCH_SIZE, t.FG_FRACTION, t.FG_THRESH, t.BG_THRESH_HI, t.BG_THRESH_LO,
                                        t.BBOX_NORMALIZE_MEANS, t.BBOX_NORMALIZE_STDS, seed=self._sample_seed + 1, num=self._num_rois,
                                        opts=ops.roi_target_opts(t.USE_GT, t.BBOX_INSIDE_WEIGHTS))
        rois, roi_scores, labels, tg, iw, ow, counts = out
        # the sampled RoIs live at ONE address for the life of the session: the tape's crop record names them, and a captured reverse
        # sweep (cfg.HIP.TRAIN_GRAPH) is only valid for the tensors it was recorded with
        rois = ops.t_copy(self._sess.buf(self._tag + "/" + name + "/rois", tuple(rois.shape)), rois)
        self._proposal_targets = dict(rois=rois, labels=labels, bbox_targets=tg, bbox_inside_weights=iw, bbox_outside_weights=ow,
                                      counts=counts)
        self._num_rois = None
        return rois, roi_scores

    def _smooth_l1_loss(self, bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0, dim=[1]):
        """network.py:264-277 -> (loss [1], d loss / d bbox_pred).  dim=[1]: mean over rows; dim=[1,2,3]: batch of 1."""
        div = float(bbox_pred.shape[0]) if list(dim) == [1] else 1.0
        return ops.smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma, div)

    def _add_losses(self, sigma_rpn=3.0):
        """network.py:279-321.  Returns the loss tensors and the seed gradients of the four network outputs."""
        A = self._num_anchors
        p, at, pt = self._predictions, self._anchor_targets, self._proposal_targets
        _, H, W, _ = p["rpn_cls_score"].shape
        rpn_ce, g_rpn_cls = ops.softmax_ce_loss(p["rpn_cls_score"], at["rpn_labels"], rpn_shape=(A, H, W))
        rpn_box, g_rpn_box = self._smooth_l1_loss(p["rpn_bbox_pred"], at["rpn_bbox_targets"], at["rpn_bbox_inside_weights"],
                                                  at["rpn_bbox_outside_weights"], sigma=sigma_rpn, dim=[1, 2, 3])
        ce, g_cls = ops.softmax_ce_loss(p["cls_score"], pt["labels"].view(-1))
        box, g_box = self._smooth_l1_loss(p["bbox_pred"], pt["bbox_targets"], pt["bbox_inside_weights"], pt["bbox_outside_weights"])
        self._losses = dict(cross_entropy=ce, loss_box=box, rpn_cross_entropy=rpn_ce, rpn_loss_box=rpn_box)
        self._loss_seeds = [(p["rpn_cls_score"], g_rpn_cls), (p["rpn_bbox_pred"], g_rpn_box), (p["cls_score"], g_cls),
                            (p["bbox_pred"], g_box)]
        return self._losses

    def _crop_images(self, bottom, rois, out, max_pool=False, bias=None, act=ACT_NONE):
        """tf.image.crop_and_resize(bottom, boxes, box_ind = rois[:,0]) (network.py:141-157) for the whole batch in one launch."""
        fs, P = float(self._feat_stride[0]), cfg.POOLING_SIZE
        self._need_f32(bottom)
        nbytes = 4 * (bottom.numel() + out.numel())
        if bias is None and act == ACT_NONE:
            self._sess.mark("op:crop_and_resize", 0, lambda: ops.crop_and_resize(bottom, rois, fs, P, max_pool=max_pool, out=out), nbytes=nbytes)
        else:
            self._sess.mark("op:crop_and_resize", 0, lambda: ops.crop_and_resize_bias_act(bottom, rois, fs, P, bias, act, out=out), nbytes=nbytes)
        return self._wrote(out)

    def _crop_pool_layer(self, bottom, rois, name):
        # network.py:141-157: 14x14 crop + 2x2 max pool (TEST: fused in one kernel)
        return self._crop_pool(bottom, rois, name, max_pool=True)

    def _crop_pool(self, bottom, rois, name, max_pool):
        """RoI pooling.  TEST: one kernel (the 2x2 max fused).  TRAIN: crop and max pool are separate tape records so that the
        reverse sweep routes the gradient through the arg-max and then through the bilinear taps."""
        P, C = cfg.POOLING_SIZE, bottom.shape[-1]
        R = rois.shape[0]
        fs = float(self._feat_stride[0])
        if self._mode != "TRAIN":
            return self._crop_images(bottom, rois, self._sess.buf(self._tag + "/" + name, (R, P, P, C)), max_pool=max_pool)
        pre = 2 * P if max_pool else P
        crop = self._sess.buf(self._tag + "/" + name + ("/crop" if max_pool else ""), (R, pre, pre, C))
        self._sess.mark("op:crop_and_resize", 0, lambda: ops.crop_and_resize(bottom, rois, fs, pre, max_pool=False, out=crop),
                        nbytes=4 * (bottom.numel() + crop.numel()))
        self._wrote(crop)
        self._tape.append(dict(kind="crop", feat=bottom, rois=rois, y=crop, stride=fs))
        if bottom.data_ptr() in self._requires_grad:
            self._requires_grad.add(crop.data_ptr())
        return self._max_pool(crop, 2, 2, name) if max_pool else crop

    def _max_pool(self, x, k, stride, name):
        """slim.max_pool2d(padding='SAME') for k == stride (pads bottom / right only, out = ceil(n / stride)); recorded in TRAIN mode."""
        N, H, W, C = x.shape
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        out = self._sess.buf(self._tag + "/" + name, (N, OH, OW, C))
        pad = (0, (OH - 1) * stride + k - H, 0, (OW - 1) * stride + k - W)
        self._sess.mark("op:maxpool", 0, lambda: ops.maxpool(x, k, stride, pad, out=out), nbytes=4 * (x.numel() + out.numel()))
        self._wrote(out)
        if self._mode == "TRAIN":
            self._tape.append(dict(kind="maxpool", x=x, y=out, k=k, stride=stride, name=name))
            if x.data_ptr() in self._requires_grad:
                self._requires_grad.add(out.data_ptr())
        return out

    def weight_decay_for(self, scope):
        """L2 coefficient of one layer's weights; None = cfg.TRAIN.WEIGHT_DECAY (network.py:303-311 arg_scope)."""
        return None

    def _region_proposal(self, net_conv, is_training, initializer=None):
        A = self._num_anchors
        rpn = self._conv(net_conv, self._scope + "/rpn_conv/3x3", 3, 1, (1, 1, 1, 1), ACT_RELU)          # :324-325
        self._act_summaries.append(rpn)
        rpn_cls_score = self._conv(rpn, self._scope + "/rpn_cls_score", 1, act=ACT_NONE)                   # :327-329
        rpn_cls_prob = self._softmax_layer(rpn_cls_score, "rpn_cls_prob")                                   # :331-334
        rpn_bbox_pred = self._conv(rpn, self._scope + "/rpn_bbox_pred", 1, act=ACT_NONE)                   # :335-337
        if is_training:                                                                                    # :338-343
            rois, roi_scores = self._proposal_layer(rpn_cls_prob, rpn_bbox_pred, "rois")
            self._anchor_target_layer(rpn_cls_score, "anchor")
            rois, _ = self._proposal_target_layer(rois, roi_scores, "rpn_rois")
        elif cfg.TEST.MODE == "nms":
            rois, _ = self._proposal_layer(rpn_cls_prob, rpn_bbox_pred, "rois")
        elif cfg.TEST.MODE == "top":
            rois, _ = self._proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, "rois")
        else:
            raise NotImplementedError
        self._predictions["rpn_cls_score"] = rpn_cls_score
        self._predictions["rpn_cls_prob"] = rpn_cls_prob
        self._predictions["rpn_bbox_pred"] = rpn_bbox_pred
        self._predictions["rois"] = rois
        return rois

    def _region_classification(self, fc7, is_training, initializer=None, initializer_bbox=None):
        R = fc7.shape[0]
        x = fc7.view(1, 1, R, fc7.shape[1])
        cls_score = self._conv(x, self._scope + "/cls_score", 1, act=ACT_NONE).view(R, self._num_classes)   # :362-366
        cls_prob = self._softmax_layer(cls_score, "cls_prob")
        affine = None
        if self._mode == "TEST":      # network.py:428-432: bbox_pred * stds + means, folded into the fc
            affine = (np.tile(np.array(cfg.TRAIN.BBOX_NORMALIZE_STDS), self._num_classes),
                      np.tile(np.array(cfg.TRAIN.BBOX_NORMALIZE_MEANS), self._num_classes))
        bbox_pred = self._conv(x, self._scope + "/bbox_pred", 1, act=ACT_NONE, out_affine=affine).view(R, 4 * self._num_classes)
        self._predictions["cls_score"] = cls_score
        self._predictions["cls_prob"] = cls_prob
        self._predictions["bbox_pred"] = bbox_pred
        return cls_prob, bbox_pred

    def _image_to_head(self, is_training, reuse=None):
        raise NotImplementedError

    def _head_to_tail(self, pool5, is_training, reuse=None):
        raise NotImplementedError

    PLAN_IMAGES = 4        # csrc/conv_igemm.hip PLAN_IMAGES: every launch-size rule is evaluated for a batch of this many images

    def _plan_rows(self, M):
        """Rows of a launch as they would be in a PLAN_IMAGES-image batch.  Which matrix pipe a GEMM runs on (>= 150 tiles: h2 / x3) and
        whether a convolution is cut along K change the bits of the result, so neither may depend on how many images share the launch:
        TEST-mode rules see per-image rows x PLAN_IMAGES whatever the batch is (the reference is strictly batch-1, lib/model/test.py:88;
        the same image must give the same tensors at batch 1, in any slot of a batch of 4, or of 8)."""
        B = getattr(self, "_plan_batch", 0)
        return M // B * self.PLAN_IMAGES if (B > 0 and M % B == 0) else M

    class _plan_context(object):
        """The batch-invariant launch plan around a graph build: TEST-mode rules (matrix pipe, split-K plan, f32 tile configuration)
        see the PER-IMAGE shape whatever the batch is -- _plan_rows here, launch context key 8 inside the library."""

        def __init__(self, net, is_training):
            self.net, self.batch = net, 0 if is_training else int(net._image.shape[0])

        def __enter__(self):
            from frcnn_hip import lib
            self.net._plan_batch = self.batch
            lib().frcnn_set_tuning(8, self.batch)        # the same rule inside the library (split-K plan of frcnn_conv2d_nhwc_ws)

        def __exit__(self, *exc):
            from frcnn_hip import lib
            lib().frcnn_set_tuning(8, 0)
            return False

    def _build_network(self, is_training=True):
        with self._plan_context(self, is_training):
            return self._build_network_impl(is_training)

    def _build_network_impl(self, is_training=True):
        self._tape = []
        self._requires_grad = set()
        self._h2_of, self._f32_missing = {}, set()      # planes are facts about THIS build's launches (buffers are reused across builds)
        net_conv = self._image_to_head(is_training)
        self._anchor_component()
        fused = self._fuse_tail_entry and not is_training and hasattr(self, "_fused_tail_entry")
        rois = self._region_proposal(net_conv, is_training)
        if cfg.POOLING_MODE != "crop":
            raise NotImplementedError
        if fused:
            fc7 = self._fused_tail_entry(net_conv, rois)                # crop commuted past the first 1x1 convs (exact algebra)
        else:
            pool5 = self._crop_pool_layer(net_conv, rois, "pool5")
            self._layers["pool5"] = pool5
            fc7 = self._head_to_tail(pool5, is_training)
        self._layers["fc7"] = fc7
        cls_prob, bbox_pred = self._region_classification(fc7, is_training)
        return rois, cls_prob, bbox_pred

    _trainable_on_device = False          # subclasses whose whole TRAIN graph has a reverse sweep (frcnn_hip/train.py) set True

    @staticmethod
    def _check_supported_cfg(mode):
        """Config keys implemented for ONE value only: refuse the others instead of silently ignoring them.  (TRAIN.USE_GT,
        RPN_CLOBBER_POSITIVES, RPN_POSITIVE_WEIGHT, the two INSIDE_WEIGHTS, TRAIN.TRUNCATED and TEST.BBOX_REG are kernel / initialiser
        arguments since round 3; POOLING_MODE is 'crop' only in the reference as well, network.py:393-396.)"""
        t = cfg.TRAIN
        fixed = [("POOLING_MODE", cfg.POOLING_MODE, "crop")]
        bad = ["%s = %r (only %r is implemented)" % (k, v, want) for k, v, want in fixed if v != want]
        if float(t.RPN_POSITIVE_WEIGHT) >= 0 and not (0.0 < float(t.RPN_POSITIVE_WEIGHT) < 1.0):
            bad.append("TRAIN.RPN_POSITIVE_WEIGHT = %r (the reference asserts 0 < p < 1, anchor_target_layer.py:103-104)" % (t.RPN_POSITIVE_WEIGHT,))
        if bad:
            raise NotImplementedError("unsupported configuration for the HIP path: " + "; ".join(bad))

    def create_architecture(self, mode, num_classes, tag=None, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
        assert tag is not None
        self._check_supported_cfg(mode)
        if mode == "TRAIN" and not self._trainable_on_device:
            raise NotImplementedError("%s: no reverse sweep for this network's TRAIN graph; TEST mode works" % type(self).__name__)
        self._tag = tag
        self._num_classes = num_classes
        self._mode = mode
        self._anchor_scales = tuple(anchor_scales)
        self._num_scales = len(anchor_scales)
        self._anchor_ratios = tuple(anchor_ratios)
        self._num_ratios = len(anchor_ratios)
        self._num_anchors = self._num_scales * self._num_ratios
        A = self._num_anchors
        self._var_specs.clear()
        self._declare_backbone()
        hc = self._head_channels()
        self._declare_conv_bias(self._scope + "/rpn_conv/3x3", 3, 3, hc, cfg.RPN_CHANNELS, 0.01)           # :239-240,324
        self._declare_conv_bias(self._scope + "/rpn_cls_score", 1, 1, cfg.RPN_CHANNELS, 2 * A, 0.01)
        self._declare_conv_bias(self._scope + "/rpn_bbox_pred", 1, 1, cfg.RPN_CHANNELS, 4 * A, 0.01)
        tc = self._tail_channels()
        self._var(self._scope + "/cls_score/weights", (tc, num_classes), "normal", 0.01)
        self._var(self._scope + "/cls_score/biases", (num_classes,), "zeros")
        self._var(self._scope + "/bbox_pred/weights", (tc, 4 * num_classes), "normal", 0.001)
        self._var(self._scope + "/bbox_pred/biases", (4 * num_classes,), "zeros")
        names = ["rois", "rpn_cls_score", "rpn_cls_prob", "rpn_bbox_pred", "cls_score", "cls_prob", "bbox_pred"]
        return {k: k for k in names}

    # ------------------------------------------------------------------ execution
    def _stage_image(self, sess, image, im_info=None):
        """[1,H,W,3] (BGR - PIXEL_MEANS, like blobs['data']) -> static device buffer [1,H,W,4]; the
        zero 4th channel lets the 7x7/3x3 stem run as a channel-folded MFMA GEMM.  im_info given: the buffer belongs to that image
        shape's scope (freed with the shape's graph); otherwise to the session."""
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        B, H, W, C = image.shape
        if im_info is None:
            buf = sess.buf(self._tag + "/image", (B, H, W, 4), zero=True)
        else:
            with self.shape_scope(sess, (B, H, W, 4), im_info):
                buf = sess.buf(self._tag + "/image", (B, H, W, 4), zero=True)
        buf[..., :C].copy_(image, non_blocking=True)
        return buf

    def graph_key(self, image_shape, im_info):
        """Everything a captured TEST-mode chain depends on: network, image shape, im_info and every switch that changes a launch."""
        c = cfg[self._mode]
        return (self._tag, self._scope, self._num_classes, self._anchor_scales, self._anchor_ratios, bool(cfg.RESNET.MAX_POOL),
                bool(cfg.USE_GPU_NMS), tuple(int(v) for v in image_shape), (float(im_info[0]), float(im_info[1])), self._mode, cfg.TEST.MODE, self._fuse_tail_entry,
                c.RPN_PRE_NMS_TOP_N, c.RPN_POST_NMS_TOP_N, c.RPN_NMS_THRESH, cfg.TEST.RPN_TOP_N, cfg.POOLING_SIZE,
                bool(cfg.HIP.WINOGRAD), int(cfg.HIP.WINOGRAD_MIN_CIN), int(cfg.HIP.WINOGRAD_M), tuple(cfg.HIP.WINOGRAD_F2_SCOPES),
                tuple(cfg.HIP.WINOGRAD_DIRECT_SCOPES), bool(cfg.HIP.WINOGRAD_7X7), bool(cfg.HIP.FUSE_TAIL_MEAN), bool(cfg.HIP.MFMA_X3), bool(cfg.USE_E2E_TF),
                bool(cfg.HIP.MFMA_H2), bool(cfg.HIP.H2_LAZY_SPLIT), int(cfg.HIP.H2_MIN_TILES), bool(cfg.HIP.H2_TRUNK_PLANES), int(cfg.HIP.H2_TILE_CFG), int(cfg.HIP.X3_TILE_CFG), bool(cfg.HIP.H2_TRAIN))

    def shape_scope(self, sess, image_shape, im_info):
        """The buffer scope of one image shape of this network (frcnn_hip/runtime.py Session.shape_scope): at most cfg.HIP.GRAPH_CACHE_SHAPES
        shapes per network tag keep their captured graph and buffers; the least recently used shape goes first.  The reference's graph takes
        [1, None, None, 3] and test_net walks an imdb of hundreds of sizes (lib/nets/network.py:386-390, lib/model/test.py:138-185)."""
        return sess.shape_scope(self.graph_key(image_shape, im_info), group=("graphs", self._tag), cap=int(cfg.HIP.GRAPH_CACHE_SHAPES))

    def forward_device(self, sess, image_d, im_info, use_graph=True):
        """Runs the network on a staged device image; fills self._predictions with DEVICE tensors.
        The whole chain is captured into one hipGraph per (tag, image shape) and replayed; graph and buffers live in the shape's scope."""
        self._sess = sess
        self._image = image_d
        self._im_info = (float(im_info[0]), float(im_info[1]), float(im_info[2]))
        sess.prepared.wait_planes()                    # a solver sharing the session re-derives filter planes on a side stream (a replay reads them)
        ops.ws_scope = self._tag                       # scratch buffers are per network tag (= per stream)
        key = self.graph_key(image_d.shape, self._im_info)
        cur = torch.cuda.current_stream(sess.device)
        with self.shape_scope(sess, image_d.shape, self._im_info):
            if not use_graph or sess.profile is not None:
                sess.flops_last_forward = 0
                self._build_network(self._mode == "TRAIN")
                return self._predictions
            if key not in sess.graphs:
                with torch.cuda.stream(sess.stream):
                    sess.stream.wait_stream(cur)
                    sess.flops_last_forward = 0
                    self._build_network(False)                   # warm-up: allocates buffers, packs weights
                    sess.stream.synchronize()
                    flops = sess.flops_last_forward
                    g = ops.Graph().capture(lambda: self._build_network(False))
                    sess.stream.synchronize()
                sess.graphs[key] = (g, dict(self._predictions), self._num_rois, flops, self._rois_per_image, image_d)
            g, preds, num, flops, per, captured = sess.graphs[key]
            self._predictions, self._num_rois, sess.flops_last_forward, self._rois_per_image = dict(preds), num, flops, per
            if captured.data_ptr() != image_d.data_ptr():
                # the graph reads the image at the address it was captured with; a caller that staged this image somewhere else (its own
                # buffer, the session-wide staging buffer vs. the shape scope's) gets it copied there first -- 4 B per pixel and channel,
                # device to device, on the stream the graph is launched on
                captured.copy_(image_d, non_blocking=True)
            g.launch()
        return self._predictions

    # only useful during testing mode
    def extract_head(self, sess, image):
        self._sess = sess
        sess.prepared.wait_planes()
        shape = (int(image.shape[0]), int(image.shape[1]), int(image.shape[2]), 4)
        with sess.shape_scope(("extract_head", self._tag, shape), group=("head", self._tag), cap=int(cfg.HIP.GRAPH_CACHE_SHAPES)):
            self._image = self._stage_image(sess, image)
            self._h2_of, self._f32_missing = {}, set()
            ops.ws_scope = self._tag
            with self._plan_context(self, False):          # the same launch plan as forward_device: the same bits for the same image
                feat = self._image_to_head(False)
            return feat.cpu().numpy()

    # only useful during testing mode
    def test_image(self, sess, image, im_info):
        """network.py:470-479: returns cls_score, cls_prob, bbox_pred, rois as numpy arrays (rows of
        the `num_rois` proposals that survived NMS)."""
        img = self._stage_image(sess, image, im_info)
        p = self.forward_device(sess, img, im_info)
        assert img.shape[0] == 1, "test_image keeps the reference's single-image contract (network.py:388); use detect_device for batches"
        n = p["rois"].shape[0] if self._num_rois is None else int(self._num_rois.item())
        return (p["cls_score"][:n].cpu().numpy(), p["cls_prob"][:n].cpu().numpy(), p["bbox_pred"][:n].cpu().numpy(),
                p["rois"][:n].cpu().numpy())

    # ------------------------------------------------------------------ training (network.py:488-516)
    def _stage_train_inputs(self, sess, blobs):
        """blobs -> the step's static input buffers: image [1,H,W,4] (like _stage_image), gt boxes in a [TRAIN_MAX_GT,5] buffer whose
        first G rows are valid (G is a launch argument of the two target layers, the only consumers), im_info as host floats (launch
        arguments; part of a recorded step's key)."""
        self._sess = sess
        self._image = self._stage_image(sess, blobs["data"])
        info = blobs["im_info"]
        self._im_info = (float(info[0]), float(info[1]), float(info[2]))
        gt = blobs["gt_boxes"]
        gt = gt if torch.is_tensor(gt) else torch.from_numpy(np.ascontiguousarray(gt, dtype=np.float32))
        G = int(gt.shape[0])
        cap = max(self.TRAIN_MAX_GT, (G + 63) // 64 * 64)
        buf = sess.buf(self._tag + "/gt_boxes", (cap, 5), zero=True)
        buf[:G].copy_(gt, non_blocking=True)
        self._gt_boxes = buf[:G]

    TRAIN_MAX_GT = 128          # rows of the static gt buffer (grows in steps of 64 for an image with more boxes: a new recorded-step key)

    def train_forward(self, sess, blobs):
        """TRAIN-mode forward + losses on the device (eager; the tape for the reverse sweep is recorded)."""
        assert self._mode == "TRAIN"
        self._stage_train_inputs(sess, blobs)
        return self._train_forward_staged(sess)

    def _train_forward_staged(self, sess):
        ops.ws_scope = self._tag
        sess.flops_last_forward = 0
        sess.prepared.enabled = bool(cfg.HIP.PREP_STREAM)
        self._build_network(True)
        return self._add_losses()

    def train_step(self, sess, blobs, train_op):
        """One SGD step.  `train_op` is the solver handle (frcnn_hip.train.TrainState with .lr set), the stand-in
        for the reference's TF train op.  Returns the five losses like network.py:488-498 (one host read-back)."""
        return tuple(float(v) for v in self.train_step_async(sess, blobs, train_op).cpu().tolist())

    def train_step_async(self, sess, blobs, train_op):
        """train_step without the host read-back: returns a DEVICE tensor [5] = (rpn_loss_cls, rpn_loss_box, loss_cls, loss_box,
        total_loss).  Nothing in the step synchronises with the host, so the launch queue stays ahead of the GPU across steps.

        cfg.HIP.TRAIN_REPLAY (default): the reference runs a step as ONE `sess.run` of a graph built once (network.py:488-498); here the
        second step of an image shape is recorded while it runs eagerly (frcnn_hip/replay.py: every C-ABI launch, event record / wait,
        tensor copy and collective of the step, streams as slots) and every later step of that shape replays the list -- the same
        launches, arguments, streams and order, hence the same bits, without the ~14 ms of Python a step costs the host."""
        assert self._mode == "TRAIN"
        with self._train_scope(sess, blobs):
            self._stage_train_inputs(sess, blobs)
            if not cfg.HIP.TRAIN_REPLAY:
                return self._train_step_body(sess, train_op, sess.buf(self._tag + "/train/losses", (5,))).clone()
            return self._train_step_replayed(sess, train_op)

    def _train_scope(self, sess, blobs):
        """A roidb's images differ in size from step to step (lib/roi_data_layer/layer.py:80-93); the step's activations, gradients, arena
        results and scratch are static per image SHAPE (a recorded step addresses them), so they live in the shape's buffer scope and at
        most cfg.HIP.TRAIN_CACHE_SHAPES shapes are kept -- the least recently used shape's buffers and recordings are dropped together."""
        d = blobs["data"]
        shape = (int(d.shape[0]), int(d.shape[1]), int(d.shape[2]), 4)
        self._train_scope_key = ("train_shape", self._tag, shape)
        return sess.shape_scope(self._train_scope_key, group=("train", self._tag), cap=int(cfg.HIP.TRAIN_CACHE_SHAPES))

    def _train_step_body(self, sess, train_op, out):
        """forward + losses + reverse sweep + solver, enqueued eagerly; the five losses into the static tensor `out`"""
        losses = self._train_forward_staged(sess)
        if not train_op.params:
            with ops.unscoped():                                                  # solver state is per session, not per image shape
                train_op.build()
                if getattr(train_op, "pending_slots", None) is not None:          # resumed run: momentum before the first update
                    train_op.import_slots(train_op.pending_slots)
                    train_op.pending_slots = None
        self.configure_train_op(train_op)
        train_op.backward(self._loss_seeds, fuse_solver=True)
        reg = train_op.regularization_value()
        parts = [losses[k].view(1) for k in ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box")]

        def assemble():
            torch.cat(parts + [reg + parts[0] + parts[1] + parts[2] + parts[3]], out=out)
        ops.host_op(assemble)
        train_op.apply(train_op.lr, getattr(train_op, "world_size", 1), getattr(train_op, "all_reduce", None))
        self._sample_seed += 2
        return out

    REPLAY_CAP = 16             # recorded steps kept per session (one per image shape; least recently used goes first)

    def _train_step_replayed(self, sess, train_op):
        import frcnn_hip
        from frcnn_hip import replay
        main = torch.cuda.current_stream(sess.device)
        hip = tuple((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in sorted(cfg.HIP.items()))
        t = cfg.TRAIN
        key = ("train_replay", self._tag, tuple(self._image.shape), self._im_info, int(self._gt_boxes.data_ptr()), float(train_op.lr),
               id(train_op), train_op.replay_signature(), hip, self._num_classes, self._anchor_scales, self._anchor_ratios, bool(cfg.RESNET.MAX_POOL),
               bool(cfg.USE_GPU_NMS), bool(cfg.USE_E2E_TF), cfg.POOLING_SIZE,
               (t.RPN_PRE_NMS_TOP_N, t.RPN_POST_NMS_TOP_N, t.RPN_NMS_THRESH, t.BATCH_SIZE, t.FG_FRACTION, t.FG_THRESH, t.BG_THRESH_HI, t.BG_THRESH_LO,
                t.RPN_BATCHSIZE, t.RPN_FG_FRACTION, t.RPN_POSITIVE_OVERLAP, t.RPN_NEGATIVE_OVERLAP, bool(t.RPN_CLOBBER_POSITIVES),
                float(t.RPN_POSITIVE_WEIGHT), tuple(t.RPN_BBOX_INSIDE_WEIGHTS), tuple(t.BBOX_INSIDE_WEIGHTS), bool(t.USE_GT),
                tuple(float(v) for v in t.BBOX_NORMALIZE_MEANS), tuple(float(v) for v in t.BBOX_NORMALIZE_STDS)))
        ent = sess.graphs.get(key)
        if ent is None:
            live = [k for k in sess.graphs if isinstance(k, tuple) and k and k[0] == "train_replay"]
            if len(live) >= self.REPLAY_CAP:
                self._drop_recording(sess, min(live, key=lambda k: sess.graphs[k]["used"]))
            ent = sess.graphs[key] = dict(seen=0, rec=None, used=0, scope=getattr(self, "_train_scope_key", None))
        self._replay_clock = getattr(self, "_replay_clock", 0) + 1
        ent["used"] = self._replay_clock
        out = sess.buf(self._tag + "/train/losses", (5,))
        rec = ent["rec"]
        if rec is not None and ent.get("gen") != sess.derived_generation():
            # the set of derived filter images changed after the recording (a TEST-mode network's first run on this session, another
            # shape's plan key): its refresh launches no longer cover the set -> this step runs eagerly (weights_changed + refresh see
            # everything) and the shape is recorded again at its next steady step
            self._forget_recording(sess, ent)
            rec = None
        if rec is not None:
            # which physical stream every helper slot runs on: inherited from the recording, or (cfg.HIP.TRAIN_PICK_STREAMS = pool size)
            # searched once per session by timing real steps (replay.StreamPicker) and then shared by every recording with the same slots
            pk = ent.get("picker")
            picked = getattr(sess, "picked_streams", None)
            if pk is None and int(cfg.HIP.TRAIN_PICK_STREAMS) > 0 and picked is None and not getattr(sess, "picking", False):
                pool = [torch.cuda.Stream(device=sess.device) for _ in range(int(cfg.HIP.TRAIN_PICK_STREAMS))]
                pk = ent["picker"] = replay.StreamPicker(rec, main, pool)
                sess.picking = True
            if pk is not None and not pk.done:
                pk.before_step(main)
                if pk.done:                                       # (nothing to try)
                    sess.picked_streams, sess.picking, sess.pick_log = list(pk.best[1:]), False, list(pk.log)
            if pk is None or pk.done:
                helpers = picked if (picked is not None and len(picked) == len(rec.streams) - 1) else list(rec.streams[1:])
                if rec.bound is None or [int(x.cuda_stream) for x in rec.bound] != [int(x.cuda_stream) for x in [main] + helpers]:
                    rec.bind([main] + helpers)
            for h, st in list(sess.prepared.readers.items()):     # a TEST-mode network on another stream read the filter images since the
                if h != int(main.cuda_stream):                    # last refresh: the list's refresh waits for `main` only, so main waits for it
                    main.wait_stream(st)
            sess.prepared.readers = {}
            rec.replay(dict(seed=self._sample_seed, gt=int(self._gt_boxes.shape[0])))
            sess.prepared.refreshed()                         # (the replayed refresh re-recorded the tier events: every reader waits again)
            self._predictions, self._losses, self._proposal_targets, self._anchor_targets = ent["views"]
            self._sample_seed += 2
            self.replay_stats["replayed"] += 1
            if pk is not None and not pk.done:
                pk.after_step(main)
                if pk.done:
                    sess.picked_streams, sess.picking, sess.pick_log = list(pk.best[1:]), False, list(pk.log)
            return out.clone()
        arena = getattr(self, "_train_arena", None)
        if arena is None or arena.sess is not sess:
            arena = self._train_arena = replay.Arena(sess, self._tag + "/train")
        arena.reset()
        prep = sess.prepared
        steady = (ent["seen"] >= 1 and bool(train_op.params) and getattr(train_op, "_sgd_table", None) is not None
                  and frcnn_hip.recorder is None
                  and (not cfg.HIP.PREP_STREAM or (prep.ready_version == prep.version and len(prep.plan) > 0 and len(prep.ready) == len(prep.plan))))
        gen_before = sess.derived_generation()
        ops.arena = arena
        try:
            if steady:
                prep.forget_waits(main)                       # the recorded step carries every tier's wait at its first use
                rec = replay.Recording(main)
                rec.vars = dict(seed=self._sample_seed, gt=int(self._gt_boxes.shape[0]))
                frcnn_hip.recorder = rec
            self._train_step_body(sess, train_op, out)
        finally:
            ops.arena = None
            frcnn_hip.recorder = None
        ent["seen"] += 1
        if steady and sess.derived_generation() == gen_before:     # (a filter image derived inline during the step = not the steady state yet)
            ent["rec"] = rec
            ent["gen"] = gen_before
            ent["views"] = (dict(self._predictions), dict(self._losses), dict(self._proposal_targets), dict(self._anchor_targets))
            self.replay_stats["recorded"] += 1
        else:
            self.replay_stats["eager"] += 1
        return out.clone()

    @staticmethod
    def _forget_recording(sess, ent):
        """Drops an entry's recording (kept: its `seen` count, so the next steady step records again).  A stream search that was running on
        this recording ends with it -- sess.picking must not stay set, or no other recording could ever start one."""
        pk = ent.pop("picker", None)
        if pk is not None and not pk.done:
            sess.picking = False
        ent["rec"] = None
        ent.pop("views", None)
        ent.pop("gen", None)

    @classmethod
    def _drop_recording(cls, sess, key):
        """LRU eviction of a recorded step (REPLAY_CAP)."""
        cls._forget_recording(sess, sess.graphs[key])
        del sess.graphs[key]

    @staticmethod
    def configure_train_op(train_op):
        """cfg.HIP -> the solver handle's switches for the reverse sweep (frcnn_hip/train.py)."""
        train_op.winograd = ((int(cfg.HIP.WINOGRAD_M), int(cfg.HIP.WINOGRAD_MIN_CIN), bool(cfg.HIP.WINOGRAD_7X7))
                             if (cfg.HIP.WINOGRAD and cfg.HIP.WINOGRAD_TRAIN) else None)
        train_op.h2_train = Network.h2_min_tiles("TRAIN") if (cfg.HIP.MFMA_H2 and cfg.HIP.H2_TRAIN) else None
        train_op.wgrad_stream = int(cfg.HIP.WGRAD_STREAM)
        train_op.wgrad_tn = bool(cfg.HIP.WGRAD_TN)
        train_op.prep_stream = bool(cfg.HIP.PREP_STREAM)
        train_op.wgrad_h2 = bool(cfg.HIP.MFMA_H2 and cfg.HIP.H2_TRAIN and cfg.HIP.WGRAD_H2)

    def train_step_no_return(self, sess, blobs, train_op):
        self.train_step(sess, blobs, train_op)

    def detect_device(self, sess, image_d, im_info, im_shape, max_per_image=100, thresh=0.0, out=None, count=None):
        """image(s) already in HBM -> final detections in HBM: forward + the whole of lib/model/test.py:95-102,
        162-180 on device.  One image: returns (dets [max_out,6], count [1]).  A batch [B,H,W,4] of same-size images:
        dets [B,max_out,6], count [B] (the dense layers run once for the batch, the per-image stages once per image)."""
        p = self.forward_device(sess, image_d, im_info)
        ops.ws_scope = self._tag
        B = image_d.shape[0]
        R, C = self._rois_per_image, self._num_classes
        # rows of the output: test.py:176-180 keeps every detection that TIES the max_per_image-th score, so the list can exceed
        # max_per_image; the default buffer holds 28 extra rows (a 128-row record), `count` reports the true number
        max_out = None if out is None else out.shape[-2]
        with self.shape_scope(sess, image_d.shape, im_info):         # (the post-processing scratch follows the proposal count of the shape's graph)
            if B > 1:
                max_out = (max_per_image + 28) if out is None else max_out
                out = sess.buf(self._tag + "/dets", (B, max_out, 6)) if out is None else out
                count = sess.buf(self._tag + "/det_count", (B,), torch.int32) if count is None else count
            return sess.mark("op:detect_post", 0, lambda: ops.detect_post(
                p["cls_prob"], p["bbox_pred"] if cfg.TEST.BBOX_REG else None, p["rois"], self._num_rois, float(im_info[2]), int(im_shape[0]), int(im_shape[1]),
                float(cfg.TEST.NMS), float(thresh), int(max_per_image), max_out=max_out, out=out, count=count, batch=B,
                rule=self._nms_rule()), nbytes=B * R * 20 * C)
