"""mobilenetv1 -- MobileNet-v1 1.0 Faster R-CNN backbone of the reference
(lib/nets/mobilenet_v1.py:63-79,114-172,214-250) on libfrcnn_hip.so.

Conv2d_0 = conv2d_same 3x3/2 (channel-folded MFMA GEMM), then 13 depthwise-separable layers:
depthwise 3x3 (VALU, bandwidth bound: frcnn_dwconv3x3_nhwc; stride 2 = explicit pad (1,1) + VALID,
mobilenet_v1.py:41-49) and pointwise 1x1 (MFMA GEMM); frozen BN (eps 1e-3, :182) folded, ReLU6
everywhere.  Layers 0-11 = stride-16 head (512 ch), layers 12-13 = per-RoI tail + spatial mean."""
import numpy as np
from frcnn_hip import ACT_RELU6, ops
from model.config import cfg
from nets.network import Network

BN_EPS = 1e-3
# (stride, depth) of the 13 DepthSepConv layers (mobilenet_v1.py:63-79; the 13th is stride 1)
_SEP = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512), (1, 512), (1, 512), (1, 512),
        (1, 1024), (1, 1024)]


class mobilenetv1(Network):
    _rgb_first_conv = "/Conv2d_0"
    dw_bn_eps = BN_EPS

    def _fix_one(self, name, value):
        # mobilenet_v1.py:266-278: tf.reverse(Conv2d_0_rgb / (255.0 / 2.0), [2]) -- the released weights expect [-1,1] inputs
        return np.ascontiguousarray((value / np.float32(255.0 / 2.0))[:, :, ::-1, :]).astype(np.float32)

    def __init__(self):
        Network.__init__(self)
        self._feat_stride = [16, ]
        self._feat_compress = [1. / float(self._feat_stride[0]), ]
        self._depth_multiplier = cfg.MOBILENET.DEPTH_MULTIPLIER
        self._scope = 'MobilenetV1'

    def _depth(self, d):
        return max(int(d * self._depth_multiplier), 8)

    def _declare_backbone(self):
        s = self._scope
        self._declare_conv_bn(s + "/Conv2d_0", 3, 3, 3, self._depth(32))
        cin = self._depth(32)
        for i, (_, d) in enumerate(_SEP, start=1):
            dw = "%s/Conv2d_%d_depthwise" % (s, i)
            self._var(dw + "/depthwise_weights", (3, 3, cin, 1), "normal", 0.47)   # He for fan_in 9
            for n_, k_ in (("gamma", "bn_gamma"), ("beta", "bn_beta"), ("moving_mean", "bn_mean"), ("moving_variance", "bn_var")):
                self._var(dw + "/BatchNorm/" + n_, (cin,), k_)
            self._declare_conv_bn("%s/Conv2d_%d_pointwise" % (s, i), 1, 1, cin, self._depth(d))
            cin = self._depth(d)

    def _head_channels(self):
        return self._depth(512)

    def _tail_channels(self):
        return self._depth(1024)

    def _dw_params(self, scope):
        key = ("dw", scope)
        sess = self._sess
        if key not in sess.packed:
            w = sess.variables[scope + "/depthwise_weights"][:, :, :, 0]          # [3,3,C]
            scale, bias = sess.fold_bn(scope, BN_EPS)
            sess.packed[key] = (sess.to_device(w * scale[None, None, :]), sess.to_device(bias))
        return sess.packed[key]

    def _separable(self, x, i, stride):
        s = self._scope
        dw_scope = "%s/Conv2d_%d_depthwise" % (s, i)
        w, b = self._dw_params(dw_scope)
        N, H, W, C = x.shape
        pad = (1, 1, 1, 1)                                   # SAME (stride 1) == explicit pad 1 (stride 2) for k=3
        OH, OW = ops.conv_out_size(H, 3, stride, 1, 1), ops.conv_out_size(W, 3, stride, 1, 1)
        out = self._sess.buf(self._tag + "/" + dw_scope, (N, OH, OW, C))
        self._need_f32(x)
        cout = self._depth(_SEP[i - 1][1])
        if C % 128 == 0 and self._h2_eligible(N * OH * OW, cout, C, 1, "%s/Conv2d_%d_pointwise" % (s, i)):
            # cfg.HIP.MFMA_H2: the depthwise kernel hands the pointwise convolution its operand planes; in TEST mode that is the only
            # reader, so no float32 tensor is written (TRAIN: the tape needs it -- both forms)
            yp = self._sess.h2_buf(self._tag + "/" + dw_scope, N * OH * OW, C)
            keep = self._mode == "TRAIN"
            self._sess.mark("op:dwconv3x3", 0, lambda: ops.dwconv3x3(x, w, b, stride, pad, ACT_RELU6, out=out, out_planes=yp, want_f32=keep),
                            nbytes=4 * (x.numel() + out.numel() * (2 if keep else 1)))
            self._wrote(out, yp, keep)
            y = out
        else:
            y = self._sess.mark("op:dwconv3x3", 0, lambda: ops.dwconv3x3(x, w, b, stride, pad, ACT_RELU6, out=out), nbytes=4 * (x.numel() + out.numel()))
            self._wrote(out)
        if self._mode == "TRAIN":
            self._tape.append(dict(kind="dwconv", scope=dw_scope, x=x, y=out, stride=stride, pad=pad, act=ACT_RELU6))
            if self.trainable_scope(dw_scope) or x.data_ptr() in self._requires_grad:
                self._requires_grad.add(out.data_ptr())
        return self._conv(y, "%s/Conv2d_%d_pointwise" % (s, i), 1, act=ACT_RELU6, bn_eps=BN_EPS)

    _trainable_on_device = True

    def create_architecture(self, mode, num_classes, tag=None, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
        if mode == "TRAIN":
            if cfg.MOBILENET.REGU_DEPTH:
                raise NotImplementedError("MOBILENET.REGU_DEPTH = True (only False is implemented: no L2 term on the depthwise filters)")
            if cfg.MOBILENET.FIXED_LAYERS < 1:
                raise NotImplementedError("MOBILENET.FIXED_LAYERS = 0: the channel-folded stem filter has no trainable master copy")
        return Network.create_architecture(self, mode, num_classes, tag, anchor_scales, anchor_ratios)

    def trainable_scope(self, scope):
        """mobilenet_v1.py:214-236: layers below cfg.MOBILENET.FIXED_LAYERS are built with is_training=False (trainable=False);
        batch-norm parameters never train (mobilenet_v1.py:176-183)."""
        tail = scope[len(self._scope):]
        if tail.startswith("/Conv2d_"):
            return int(tail[len("/Conv2d_"):].split("_")[0]) >= cfg.MOBILENET.FIXED_LAYERS
        return True

    def weight_decay_for(self, scope):
        # mobilenet_v1.py:186: the backbone's own regulariser coefficient; RPN / heads keep cfg.TRAIN.WEIGHT_DECAY
        return float(cfg.MOBILENET.WEIGHT_DECAY) if scope[len(self._scope):].startswith("/Conv2d_") else None

    def _image_to_head(self, is_training, reuse=None):
        assert (0 <= cfg.MOBILENET.FIXED_LAYERS <= 12)
        net = self._conv(self._image, self._scope + "/Conv2d_0", 3, 2, (1, 1, 1, 1), act=ACT_RELU6, bn_eps=BN_EPS,
                         fold_w=True, real_cin=3)
        for i in range(1, 12):
            net = self._separable(net, i, _SEP[i - 1][0])
        self._act_summaries.append(net)
        self._layers['head'] = net
        return net

    def _head_to_tail(self, pool5, is_training, reuse=None):
        net = pool5
        for i in (12, 13):
            net = self._separable(net, i, _SEP[i - 1][0])
        out = self._sess.buf(self._tag + "/fc7", (net.shape[0], net.shape[-1]))
        res = self._sess.mark("op:spatial_mean", 0, lambda: ops.spatial_mean(net, out=out), nbytes=4 * (net.numel() + out.numel()))
        if self._mode == "TRAIN":
            self._tape.append(dict(kind="mean", x=net, y=res, name=self._scope + "/fc7_mean"))
            if net.data_ptr() in self._requires_grad:
                self._requires_grad.add(res.data_ptr())
        return res
