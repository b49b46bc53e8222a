"""vgg16 -- the VGG16 Faster R-CNN backbone of the reference (lib/nets/vgg16.py:20-60) on
libfrcnn_hip.so: 13 3x3 SAME conv+bias+ReLU layers (implicit GEMM on the f32 MFMA pipe), four 2x2/2
SAME max pools -> stride-16 conv5_3; RoI pooling = 14x14 crop + 2x2 max (network.py:141-157, fused in
one kernel in TEST mode); fc6/fc7 as 1x1 convs over the NHWC-flattened [R,7*7*512] crops; TRAIN adds dropout6/7
and records pools / crops / dropouts on the tape for the reverse sweep (frcnn_hip/train.py)."""
import numpy as np
from frcnn_hip import ACT_RELU, ops
from nets.network import Network

_CFG = [("conv1", 2, 64), ("conv2", 2, 128), ("conv3", 3, 256), ("conv4", 3, 512), ("conv5", 3, 512)]


class vgg16(Network):
    _rgb_first_conv = "/conv1/conv1_1"

    def _variables_to_fix_names(self):
        # vgg16.py:62-79: the ImageNet checkpoint stores fc6 / fc7 as conv filters [7,7,512,4096] / [1,1,4096,4096]
        return [self._scope + t + "/weights" for t in ("/fc6", "/fc7", "/conv1/conv1_1")]

    def _fix_one(self, name, value):
        if name.endswith("/fc6/weights") or name.endswith("/fc7/weights"):        # tf.reshape(fc6_conv, fc6.get_shape()), :93-96
            return np.ascontiguousarray(value).reshape(self._var_specs[name].shape)
        return np.ascontiguousarray(value[:, :, ::-1, :])                         # tf.reverse(conv1_rgb, [2]), :97-98

    def __init__(self):
        Network.__init__(self)
        self._feat_stride = [16, ]
        self._feat_compress = [1. / float(self._feat_stride[0]), ]
        self._scope = 'vgg_16'

    def _declare_backbone(self):
        cin = 3
        for name, reps, depth in _CFG:
            for r in range(1, reps + 1):
                scope = "%s/%s/%s_%d" % (self._scope, name, name, r)
                self._var(scope + "/weights", (3, 3, cin, depth), "he")
                self._var(scope + "/biases", (depth,), "zeros")
                cin = depth
        self._var(self._scope + "/fc6/weights", (7 * 7 * 512, 4096), "he")
        self._var(self._scope + "/fc6/biases", (4096,), "zeros")
        self._var(self._scope + "/fc7/weights", (4096, 4096), "he")
        self._var(self._scope + "/fc7/biases", (4096,), "zeros")

    def _head_channels(self):
        return 512

    def _tail_channels(self):
        return 4096

    _trainable_on_device = True

    def trainable_scope(self, scope):
        """vgg16.py:29-33: conv1_x and conv2_x are created with trainable=False; everything else trains (weights and biases)."""
        tail = scope[len(self._scope):]
        return not (tail.startswith("/conv1/") or tail.startswith("/conv2/"))

    def _image_to_head(self, is_training, reuse=None):
        net = self._image                                   # [1,H,W,4]: first conv runs channel-folded
        first = True
        for bi, (name, reps, depth) in enumerate(_CFG):
            for r in range(1, reps + 1):
                scope = "%s/%s/%s_%d" % (self._scope, name, name, r)
                net = self._conv(net, scope, 3, 1, (1, 1, 1, 1), ACT_RELU, fold_w=first, real_cin=3 if first else None)
                first = False
            if bi < 4:                                      # pool1..pool4, 'SAME': out = ceil(n/2)
                net = self._max_pool(net, 2, 2, "pool%d" % (bi + 1))
        self._act_summaries.append(net)
        self._layers['head'] = net
        return net

    def _dropout(self, x, layer):
        """slim.dropout(keep_prob=0.5, is_training=True) (vgg16.py:52-58); the mask is a function of (step seed, layer, element)."""
        seed = (int(self._sample_seed) << 8) | int(layer)
        out = self._sess.buf(self._tag + "/dropout%d" % layer, tuple(x.shape))
        self._sess.mark("op:dropout", 0, lambda: ops.dropout(x, seed, 0.5, out=out, step_mult=256), nbytes=8 * x.numel())
        self._tape.append(dict(kind="dropout", x=x, y=out, seed=seed, keep=0.5, name="dropout%d" % layer))
        if x.data_ptr() in self._requires_grad:
            self._requires_grad.add(out.data_ptr())
        return out

    def _head_to_tail(self, pool5, is_training, reuse=None):
        # slim.flatten is NHWC order, so fc6's [25088,4096] matrix is a 1x1 filter over the flattened [R,7*7*512] crops (the k order
        # (h, w, c) is also the slab order of a 7x7 VALID convolution: same sums)
        R = pool5.shape[0]
        flat = pool5.view(1, 1, R, pool5[0].numel())
        fc6 = self._conv(flat, self._scope + "/fc6", 1, act=ACT_RELU)                      # [1,1,R,4096]
        if is_training:
            fc6 = self._dropout(fc6, 6)
        fc7 = self._conv(fc6, self._scope + "/fc7", 1, act=ACT_RELU)
        if is_training:
            fc7 = self._dropout(fc7, 7)
        return fc7.view(R, 4096)
