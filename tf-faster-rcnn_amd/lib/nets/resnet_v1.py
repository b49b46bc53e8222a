"""resnetv1 -- ResNet-50/101/152 Faster R-CNN backbone of the reference (lib/nets/resnet_v1.py) on
libfrcnn_hip.so: every convolution is an implicit-GEMM launch on the f32 MFMA pipe with the frozen
batch norm (resnet_v1.py:22-44: is_training False, trainable False) folded into filter + bias, and
ReLU / residual add / `subsample` shortcut fused into the epilogue.

slim semantics restated (third-party):
  * conv1 = resnet_utils.conv2d_same(64, 7, stride 2) (pad 3/3 then VALID), pad 1 + 3x3/2 VALID pool
    (resnet_v1.py:80-86);
  * bottleneck_v1: shortcut = 1x1 conv (BN, no activation) when depth changes, else
    subsample(inputs, stride) (= every stride-th pixel); conv1 1x1 -> conv2 3x3 conv2d_same(stride)
    -> conv3 1x1 (BN, no activation); out = relu(shortcut + residual);
  * resnet_v1_block: the stride sits on the LAST unit of a block;
  * blocks 1-3 form the stride-16 head (block3 stride 1), block4 (stride 1) runs per RoI on the
    7x7 crops followed by a spatial mean (resnet_v1.py:88-125).
"""
from frcnn_hip import ACT_NONE, ACT_RELU, ops
from model.config import cfg
from nets.network import Network

BN_EPS = 1e-5      # resnet_v1.py:24


def _same_pad(k, stride):
    """resnet_utils.conv2d_same: stride 1 -> 'SAME'; stride > 1 -> explicit pad (k-1)//2, rest."""
    total = k - 1
    beg = total // 2
    return (beg, total - beg, beg, total - beg)


class resnetv1(Network):
    _rgb_first_conv = "/conv1"
    _trainable_on_device = True
    def __init__(self, num_layers=50):
        Network.__init__(self)
        self._feat_stride = [16, ]
        self._feat_compress = [1. / float(self._feat_stride[0]), ]
        self._num_layers = num_layers
        self._scope = 'resnet_v1_%d' % num_layers
        self._decide_blocks()

    def _decide_blocks(self):
        # (name, base_depth, num_units, stride)  (resnet_v1.py:127-152)
        units = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}.get(self._num_layers)
        if units is None:
            raise NotImplementedError
        self._blocks = [("block1", 64, units[0], 2), ("block2", 128, units[1], 2),
                        ("block3", 256, units[2], 1),          # stride 1 for the last conv4 layer
                        ("block4", 512, units[3], 1)]

    # ---- variables (slim names) ------------------------------------------------------------------
    def _declare_backbone(self):
        s = self._scope
        self._declare_conv_bn(s + "/conv1", 7, 7, 3, 64)
        cin = 64
        for name, base, n_units, _ in self._blocks:
            depth = base * 4
            for u in range(1, n_units + 1):
                p = "%s/%s/unit_%d/bottleneck_v1" % (s, name, u)
                if cin != depth:
                    self._declare_conv_bn(p + "/shortcut", 1, 1, cin, depth)
                self._declare_conv_bn(p + "/conv1", 1, 1, cin, base)
                self._declare_conv_bn(p + "/conv2", 3, 3, base, base)
                self._declare_conv_bn(p + "/conv3", 1, 1, base, depth, residual_branch_end=True)
                cin = depth

    def _head_channels(self):
        return self._blocks[2][1] * 4

    def _tail_channels(self):
        return self._blocks[3][1] * 4

    # ---- graph -----------------------------------------------------------------------------------
    def _bottleneck(self, x, prefix, base, stride, mean_rows=0, emit_out=True, f32_out=True):
        """mean_rows > 0 (TEST mode, last unit of the tail): returns mean over every `mean_rows` consecutive pixels of the unit's
        output instead of the output itself (conv3 + residual + ReLU + reduce_mean in one kernel).
        cfg.HIP.MFMA_H2 (TEST mode): conv2 hands conv3 its operand planes only (nothing else reads conv2's float32 result); conv3
        emits float32 (the next residual) AND the planes the next unit's conv1 / shortcut read (emit_out = False: a consumer that
        is not a GEMM follows, e.g. the spatial mean); f32_out = False (cfg.HIP.H2_TRUNK_PLANES): the next unit reads its input AND its
        residual from those planes, so conv3 does not write the float32 tensor at all."""
        depth = base * 4
        cin = x.shape[-1]
        if cin != depth:
            shortcut = self._conv(x, prefix + "/shortcut", 1, stride, act=ACT_NONE, bn_eps=BN_EPS)
            res_stride = 1
        else:
            shortcut, res_stride = x, stride                # slim `subsample`: fused into conv3's epilogue
        r = self._conv(x, prefix + "/conv1", 1, 1, act=ACT_RELU, bn_eps=BN_EPS)
        pad = (1, 1, 1, 1) if stride == 1 else _same_pad(3, stride)
        N, H, W, _ = r.shape
        M3 = N * ops.conv_out_size(H, 3, stride, pad[0], pad[1]) * ops.conv_out_size(W, 3, stride, pad[2], pad[3])
        if mean_rows and res_stride == 1 and stride == 1 and self._mean_fusable(M3, depth, base, prefix + "/conv3", mean_rows):
            r = self._conv(r, prefix + "/conv2", 3, stride, pad, act=ACT_RELU, bn_eps=BN_EPS, emit_h2=True, want_f32=False)
            if self._h2_input(r) is not None:      # planes from conv2's output transform, or (direct conv2) a lazy split of its f32 result
                return self._conv1x1_mean(r, prefix + "/conv3", mean_rows, act=ACT_RELU, bn_eps=BN_EPS, residual=shortcut)
            # conv2 took the direct kernel (WINOGRAD_DIRECT_SCOPES / WINOGRAD_MIN_CIN) and H2_LAZY_SPLIT is off: no planes exist, so the
            # unfused pair -- conv3 on whatever pipe its float32 input allows, then the spatial mean
            x = self._conv(r, prefix + "/conv3", 1, 1, act=ACT_RELU, bn_eps=BN_EPS, residual=shortcut, res_stride=1)
            out = self._sess.buf(self._tag + "/fc7", (x.shape[0], x.shape[-1]))
            return self._sess.mark("op:spatial_mean", 0, lambda: ops.spatial_mean(x, out=out), nbytes=4 * (x.numel() + out.numel()))
        c3_h2 = res_stride == 1 and self._h2_eligible(M3, depth, base, 1, prefix + "/conv3")
        r = self._conv(r, prefix + "/conv2", 3, stride, pad, act=ACT_RELU, bn_eps=BN_EPS, emit_h2=c3_h2, want_f32=not c3_h2)
        return self._conv(r, prefix + "/conv3", 1, 1, act=ACT_RELU, bn_eps=BN_EPS, residual=shortcut, res_stride=res_stride,
                          emit_h2=emit_out, want_f32=f32_out)

    def _trunk_planes_only(self, rows, base, next_stride, next_prefix=None):
        """cfg.HIP.H2_TRUNK_PLANES: may a unit hand its output to the NEXT unit of the same block (identity shortcut) as operand
        planes only?  Yes when that unit's conv1 reads planes (h2-eligible), its conv3 takes the residual from planes (h2-eligible,
        stride 1) -- then nothing reads the float32 tensor.  The planes carry >= 22 significant bits (csrc/gemm_h2.hip)."""
        depth = 4 * base
        return (bool(cfg.HIP.H2_TRUNK_PLANES) and next_stride == 1
                and self._h2_eligible(rows, base, depth, 1, None if next_prefix is None else next_prefix + "/conv1")
                and self._h2_eligible(rows, depth, base, 1, None if next_prefix is None else next_prefix + "/conv3"))

    def _run_blocks(self, x, blocks, emit_last=True):
        for bi, (name, base, n_units, stride) in enumerate(blocks):
            for u in range(1, n_units + 1):
                s_u = stride if u == n_units else 1
                N, H, W, _ = x.shape
                rows = N * ops.conv_out_size(H, 3, s_u, 1, 1) * ops.conv_out_size(W, 3, s_u, 1, 1) if s_u > 1 else N * H * W
                planes_only = u < n_units and self._trunk_planes_only(rows, base, stride if u + 1 == n_units else 1,
                                                                       "%s/%s/unit_%d/bottleneck_v1" % (self._scope, name, u + 1))
                x = self._bottleneck(x, "%s/%s/unit_%d/bottleneck_v1" % (self._scope, name, u), base, s_u,
                                     emit_out=emit_last or u < n_units or bi + 1 < len(blocks), f32_out=not planes_only)
        return x

    def _crop_pool_layer(self, bottom, rois, name):
        # resnet_v1.py:55-76: direct 7x7 crop unless RESNET.MAX_POOL
        return self._crop_pool(bottom, rois, name, max_pool=bool(cfg.RESNET.MAX_POOL))

    def _build_base(self):
        # resnet_v1.py:80-86.  The image buffer is [1,H,W,4]: 7x7x3 stem as a channel-folded GEMM.
        net = self._conv(self._image, self._scope + "/conv1", 7, 2, _same_pad(7, 2), act=ACT_RELU, bn_eps=BN_EPS,
                         fold_w=True, real_cin=3)
        N, H, W, C = net.shape
        out = self._sess.buf(self._tag + "/pool1", (N, ops.conv_out_size(H, 3, 2, 1, 1), ops.conv_out_size(W, 3, 2, 1, 1), C))
        self._sess.mark("op:maxpool", 0, lambda: ops.maxpool(net, 3, 2, (1, 1, 1, 1), out=out), nbytes=4 * (net.numel() + out.numel()))
        return self._wrote(out)

    def _image_to_head(self, is_training, reuse=None):
        assert (0 <= cfg.RESNET.FIXED_BLOCKS <= 3)
        net_conv = self._run_blocks(self._build_base(), self._blocks[0:3])
        self._act_summaries.append(net_conv)
        self._layers['head'] = net_conv
        return net_conv

    def _fused_tail_maps(self, net_conv):
        """The two 1x1 convolutions at the entry of block4/unit_1 on the whole feature map (no bias: it is added after the crop).
        They depend on the head only."""
        name = self._blocks[-1][0]
        prefix = "%s/%s/unit_1/bottleneck_v1" % (self._scope, name)
        sc_map = self._conv(net_conv, prefix + "/shortcut", 1, 1, act=ACT_NONE, bn_eps=BN_EPS, no_bias=True)
        c1_map = self._conv(net_conv, prefix + "/conv1", 1, 1, act=ACT_NONE, bn_eps=BN_EPS, no_bias=True)
        return sc_map, c1_map

    def _fused_tail_entry(self, net_conv, rois):
        """TEST-mode restructuring of the entry of block4/unit_1 (same result up to f32 rounding): the two 1x1
        convolutions that read the RoI crops (projection shortcut, conv1) are linear per pixel and the bilinear
        crop is linear too, so they run ONCE on the 38x63 map and their outputs are cropped, instead of cropping
        first and convolving 14 700 RoI pixels:  conv1x1(crop(F)) + b == crop(conv1x1(F)) + b.
        Saves 64.5 of the 622.3 GFLOP per image (shortcut 61.7 -> 10.0, conv1 15.4 -> 2.5) and the pool5 tensor."""
        sess, P = self._sess, cfg.POOLING_SIZE
        name, base, n_units, stride = self._blocks[-1]
        prefix = "%s/%s/unit_1/bottleneck_v1" % (self._scope, name)
        R = rois.shape[0]
        sc_map, c1_map = self._fused_tail_maps(net_conv)
        b_sc = sess.conv_info[prefix + "/shortcut"]["b"]
        b_c1 = sess.conv_info[prefix + "/conv1"]["b"]
        fs = float(self._feat_stride[0])
        sc_out = sess.buf(self._tag + "/" + prefix + "/shortcut_crop", (R, P, P, sc_map.shape[-1]))
        c1_out = sess.buf(self._tag + "/" + prefix + "/conv1_crop", (R, P, P, c1_map.shape[-1]))
        shortcut = self._crop_images(sc_map, rois, sc_out, bias=b_sc, act=ACT_NONE)
        r = self._crop_images(c1_map, rois, c1_out, bias=b_c1, act=ACT_RELU)
        c3_h2 = self._h2_eligible(R * P * P, 4 * base, base, 1, prefix + "/conv3")
        r = self._conv(r, prefix + "/conv2", 3, 1, (1, 1, 1, 1), act=ACT_RELU, bn_eps=BN_EPS, emit_h2=c3_h2, want_f32=not c3_h2)
        x = self._conv(r, prefix + "/conv3", 1, 1, act=ACT_RELU, bn_eps=BN_EPS, residual=shortcut, res_stride=1, emit_h2=n_units >= 2,
                       want_f32=not (n_units >= 2 and self._trunk_planes_only(R * P * P, base, stride if n_units == 2 else 1,
                                                                              "%s/%s/unit_2/bottleneck_v1" % (self._scope, name))))
        fused = stride == 1 and n_units >= 2 and self._mean_fusable(R * P * P, 4 * base, base, "%s/%s/unit_%d/bottleneck_v1/conv3" % (self._scope, name, n_units), P * P)
        for u in range(2, n_units + 1):
            x = self._bottleneck(x, "%s/%s/unit_%d/bottleneck_v1" % (self._scope, name, u), base, stride if u == n_units else 1,
                                 mean_rows=P * P if (fused and u == n_units) else 0, emit_out=u < n_units,
                                 f32_out=not (u < n_units and self._trunk_planes_only(R * P * P, base, stride if u + 1 == n_units else 1,
                                                                                      "%s/%s/unit_%d/bottleneck_v1" % (self._scope, name, u + 1))))
        if fused:
            return x                                          # [R, 2048]: the mean came out of the last conv3's epilogue
        out = sess.buf(self._tag + "/fc7", (x.shape[0], x.shape[-1]))
        return sess.mark("op:spatial_mean", 0, lambda: ops.spatial_mean(x, out=out), nbytes=4 * (x.numel() + out.numel()))

    def _head_to_tail(self, pool5, is_training, reuse=None):
        name, base, n_units, stride = self._blocks[-1]
        if stride == 1 and self._mean_fusable(pool5.shape[0] * pool5.shape[1] * pool5.shape[2], 4 * base, base,
                                              "%s/%s/unit_%d/bottleneck_v1/conv3" % (self._scope, name, n_units), pool5.shape[1] * pool5.shape[2]):
            x = pool5
            hw = pool5.shape[1] * pool5.shape[2]
            for u in range(1, n_units + 1):
                x = self._bottleneck(x, "%s/%s/unit_%d/bottleneck_v1" % (self._scope, name, u), base, 1, mean_rows=hw if u == n_units else 0)
            return x
        fc7 = self._run_blocks(pool5, self._blocks[-1:], emit_last=False)
        # average pooling done by reduce_mean (resnet_v1.py:124)
        out = self._sess.buf(self._tag + "/fc7", (fc7.shape[0], fc7.shape[-1]))
        res = self._sess.mark("op:spatial_mean", 0, lambda: ops.spatial_mean(fc7, out=out), nbytes=4 * (fc7.numel() + out.numel()))
        if self._mode == "TRAIN":
            self._tape.append(dict(kind="mean", x=fc7, y=res, name=self._scope + "/fc7_mean"))
            if fc7.data_ptr() in self._requires_grad:
                self._requires_grad.add(res.data_ptr())
        return res

    def trainable_scope(self, scope):
        """resnet_v1.py:88-113: conv1 and the first cfg.RESNET.FIXED_BLOCKS blocks are frozen; everything else trains
        (BN statistics and affine parameters are always frozen)."""
        fixed = ["/conv1"] + ["/block%d/" % b for b in range(1, cfg.RESNET.FIXED_BLOCKS + 1)]
        tail = scope[len(self._scope):]
        return not any(tail.startswith(f) for f in fixed)
