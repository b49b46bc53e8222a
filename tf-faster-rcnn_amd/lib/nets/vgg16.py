"""vgg16 -- the VGG16 Faster R-CNN backbone of the reference (lib/nets/vgg16.py:20-60) on
libfrcnn_hip.so: 13 3x3 SAME conv+bias+ReLU layers (implicit GEMM on the f32 MFMA pipe), four 2x2/2
SAME max pools -> stride-16 conv5_3; RoI pooling = 14x14 crop + 2x2 max (network.py:141-157, fused in
one kernel); fc6/fc7 as a 7x7 VALID conv and a 1x1 conv over the [R,7,7,512] crops (slim.flatten is
NHWC order, so fc6's [25088,4096] matrix IS the HWIO filter [7,7,512,4096])."""
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

    def _image_to_head(self, is_training, reuse=None):
        net = self._image                                   # [1,H,W,4]: first conv runs channel-folded
        first = True
        for bi, (name, reps, depth) in enumerate(_CFG):
            for r in range(1, reps + 1):
                scope = "%s/%s/%s_%d" % (self._scope, name, name, r)
                net = self._conv(net, scope, 3, 1, (1, 1, 1, 1), ACT_RELU, fold_w=first, real_cin=3 if first else None)
                first = False
            if bi < 4:                                      # pool1..pool4, 'SAME': out = ceil(n/2)
                N, H, W, C = net.shape
                OH, OW = (H + 1) // 2, (W + 1) // 2
                out = self._sess.buf(self._tag + "/pool%d" % (bi + 1), (N, OH, OW, C))
                x = net
                net = self._sess.mark("op:maxpool", 0, lambda: ops.maxpool(x, 2, 2, (0, H % 2, 0, W % 2), out=out), nbytes=4 * (x.numel() + out.numel()))
        self._act_summaries.append(net)
        self._layers['head'] = net
        return net

    def _head_to_tail(self, pool5, is_training, reuse=None):
        if is_training:
            raise NotImplementedError("dropout6/7 (vgg16.py:52-58) belong to the training path")
        R = pool5.shape[0]
        w6 = self._sess.variables[self._scope + "/fc6/weights"]
        if w6.ndim == 2:                                    # [25088,4096] -> HWIO [7,7,512,4096]
            self._sess.variables[self._scope + "/fc6/weights"] = w6.reshape(7, 7, 512, 4096)
        fc6 = self._conv(pool5, self._scope + "/fc6", 7, 1, (0, 0, 0, 0), ACT_RELU)        # [R,1,1,4096]
        fc7 = self._conv(fc6.view(1, 1, R, 4096), self._scope + "/fc7", 1, act=ACT_RELU)
        return fc7.view(R, 4096)
