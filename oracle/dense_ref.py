"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the dense (TensorFlow/slim) part of the
reference graph for ResNet-v1 Faster R-CNN.  PARITY UNPINNED for these ops: TensorFlow r1.2 +
tf.contrib.slim (README.md:56; pip-unpinned in docker/Dockerfile.cuda-8.0) is a third-party
dependency that is neither vendored in /root/reference nor installable here, and the reference has
no test that pins its arithmetic.  This file restates the published semantics of the
call sites:

  lib/nets/resnet_v1.py:80-86   conv1 = conv2d_same(64,7,stride 2) + BN + ReLU, pad 1, 3x3/2 VALID pool
  lib/nets/resnet_v1.py:88-125  blocks 1-3 (head, block3 stride 1), block4 per RoI + reduce_mean
  slim bottleneck_v1            shortcut 1x1 conv(BN, no act) if depth changes else subsample;
                                conv1 1x1 -> conv2 3x3 conv2d_same(stride) -> conv3 1x1 (no act); relu(sum)
  slim batch_norm (frozen)      gamma * (x - moving_mean) / sqrt(moving_variance + 1e-5) + beta
  lib/nets/network.py:323-378   RPN 3x3 (SAME, bias, ReLU), 1x1 heads, pair softmax; fc cls/bbox heads
  lib/nets/network.py:428-432   test-time bbox_pred * stds + means

It is the reference for the HIP dense kernels (float64 mode) and the host-CPU baseline that
bench.py times (float32 mode, all cores): used ONLY by tests/, smoke() and bench.py's cpu_baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

import frcnn_oracle as ora

UNITS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


class DenseRef(object):
    def __init__(self, variables, num_layers=101, num_classes=21, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2),
                 dtype=torch.float64):
        self.v = variables
        self.scope = "resnet_v1_%d" % num_layers
        u = UNITS[num_layers]
        self.blocks = [("block1", 64, u[0], 2), ("block2", 128, u[1], 2), ("block3", 256, u[2], 1), ("block4", 512, u[3], 1)]
        self.C = num_classes
        self.scales, self.ratios = tuple(anchor_scales), tuple(anchor_ratios)
        self.A = len(anchor_scales) * len(anchor_ratios)
        self.dtype = dtype
        self._cache = {}

    # weights as torch OIHW, cached
    def w(self, name):
        if name not in self._cache:
            a = self.v[name]
            if a.ndim == 4:
                a = np.transpose(a, (3, 2, 0, 1))
            self._cache[name] = _t(a, self.dtype)
        return self._cache[name]

    def conv_same(self, x, scope, k, stride):
        """slim conv2d SAME for stride 1; resnet_utils.conv2d_same (explicit pad + VALID) otherwise."""
        total = k - 1
        beg = total // 2
        x = F.pad(x, (beg, total - beg, beg, total - beg))
        return F.conv2d(x, self.w(scope + "/weights"), stride=stride)

    def bn(self, x, scope, eps=1e-5):
        g, b = self.w(scope + "/BatchNorm/gamma"), self.w(scope + "/BatchNorm/beta")
        m, var = self.w(scope + "/BatchNorm/moving_mean"), self.w(scope + "/BatchNorm/moving_variance")
        return (x - m[None, :, None, None]) * (g / torch.sqrt(var + eps))[None, :, None, None] + b[None, :, None, None]

    def bottleneck(self, x, p, base, stride):
        depth = base * 4
        if x.shape[1] != depth:
            sc = self.bn(F.conv2d(x, self.w(p + "/shortcut/weights"), stride=stride), p + "/shortcut")
        else:
            sc = x if stride == 1 else x[:, :, ::stride, ::stride]          # slim subsample (1x1 max pool, stride)
        r = F.relu(self.bn(F.conv2d(x, self.w(p + "/conv1/weights")), p + "/conv1"))
        r = F.relu(self.bn(self.conv_same(r, p + "/conv2", 3, stride), p + "/conv2"))
        r = self.bn(F.conv2d(r, self.w(p + "/conv3/weights")), p + "/conv3")
        return F.relu(sc + r)

    def run_blocks(self, x, blocks):
        for name, base, n, stride in blocks:
            for u in range(1, n + 1):
                x = self.bottleneck(x, "%s/%s/unit_%d/bottleneck_v1" % (self.scope, name, u), base, stride if u == n else 1)
        return x

    def head(self, image_nhwc):
        x = _t(image_nhwc, self.dtype).permute(0, 3, 1, 2)
        x = F.relu(self.bn(self.conv_same(x, self.scope + "/conv1", 7, 2), self.scope + "/conv1"))
        x = F.max_pool2d(F.pad(x, (1, 1, 1, 1)), 3, 2)
        return self.run_blocks(x, self.blocks[:3])                            # NCHW

    def rpn(self, feat):
        s = self.scope
        r = F.relu(F.conv2d(feat, self.w(s + "/rpn_conv/3x3/weights"), self.w(s + "/rpn_conv/3x3/biases"), padding=1))
        score = F.conv2d(r, self.w(s + "/rpn_cls_score/weights"), self.w(s + "/rpn_cls_score/biases"))
        bbox = F.conv2d(r, self.w(s + "/rpn_bbox_pred/weights"), self.w(s + "/rpn_bbox_pred/biases"))
        A = self.A
        pair = torch.softmax(torch.stack([score[:, :A], score[:, A:]]), dim=0)  # channels (a, A+a) (network.py:68-86)
        prob = torch.cat([pair[0], pair[1]], dim=1)
        to_nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().numpy()
        return to_nhwc(score), to_nhwc(prob), to_nhwc(bbox)

    def tail(self, pool5_nhwc):
        x = _t(pool5_nhwc, self.dtype).permute(0, 3, 1, 2)
        return self.run_blocks(x, self.blocks[3:]).mean(dim=(2, 3))           # fc7 [R, 2048]

    def classify(self, fc7, test_mode=True):
        s = self.scope
        cls_score = fc7 @ self.w(s + "/cls_score/weights") + self.w(s + "/cls_score/biases")
        cls_prob = torch.softmax(cls_score, dim=1)
        bbox = fc7 @ self.w(s + "/bbox_pred/weights") + self.w(s + "/bbox_pred/biases")
        if test_mode:
            bbox = bbox * _t(np.tile((0.1, 0.1, 0.2, 0.2), self.C), self.dtype) + _t(np.tile((0.0,) * 4, self.C), self.dtype)
        return cls_score.numpy(), cls_prob.numpy(), bbox.numpy()

    max_pool_crop = False          # resnet: direct 7x7 crop (resnet_v1.py:55-76); vgg/mobilenet: 14x14 + 2x2 max

    def test_image(self, image_nhwc, im_info, rois=None, pre=6000, post=300, thr=0.7, pool=7):
        """Full reference forward (network.py:233-262, TEST, MODE nms).  If `rois` is given the RoI
        stage uses THEM (so a tail comparison is not confounded by a proposal that flipped on a
        1e-7 score difference)."""
        feat = self.head(image_nhwc)
        score, prob, bbox = self.rpn(feat)
        H, W = feat.shape[2], feat.shape[3]
        if rois is None:
            anchors, _ = ora.generate_anchors_pre(H, W, 16, self.scales, self.ratios)
            rois, _ = ora.proposal_layer(prob.astype(np.float32), bbox.astype(np.float32), np.asarray(im_info, dtype=np.float32),
                                         "TEST", [16], anchors, self.A, pre_nms_topN=pre, post_nms_topN=post, nms_thresh=thr)
        feat_nhwc = feat.permute(0, 2, 3, 1).contiguous().numpy()
        pool5 = ora.crop_and_resize(feat_nhwc[0].astype(np.float32), rois.astype(np.float32), 16.0, pool,
                                    max_pool=self.max_pool_crop)
        fc7 = self.tail(pool5)
        cls_score, cls_prob, bbox_pred = self.classify(fc7)
        return dict(head=feat_nhwc, rpn_cls_score=score, rpn_cls_prob=prob, rpn_bbox_pred=bbox, rois=rois, pool5=pool5,
                    fc7=fc7.numpy(), cls_score=cls_score, cls_prob=cls_prob, bbox_pred=bbox_pred)


def crop_and_resize_torch(feat, rois, stride, P):
    """Differentiable (w.r.t. feat) torch statement of tf.image.crop_and_resize as the reference calls it
    (nets/resnet_v1.py:55-76): sample positions in float32 exactly like oracle_c.c, bilinear taps by gather."""
    _, C, H, W = feat.shape
    r = np.asarray(rois, dtype=np.float32)
    f32 = np.float32
    height, width = (f32(H) - f32(1)) * f32(stride), (f32(W) - f32(1)) * f32(stride)
    x1, y1, x2, y2 = r[:, 1] / width, r[:, 2] / height, r[:, 3] / width, r[:, 4] / height
    hs = (y2 - y1) * f32(H - 1) / f32(P - 1)
    ws = (x2 - x1) * f32(W - 1) / f32(P - 1)
    g = np.arange(P, dtype=np.float32)
    in_y = (y1 * f32(H - 1))[:, None] + g[None, :] * hs[:, None]             # [R,P]
    in_x = (x1 * f32(W - 1))[:, None] + g[None, :] * ws[:, None]
    vy = (in_y >= 0) & (in_y <= f32(H - 1))
    vx = (in_x >= 0) & (in_x <= f32(W - 1))
    top, bot = np.floor(in_y).astype(np.int64).clip(0, H - 1), np.ceil(in_y).astype(np.int64).clip(0, H - 1)
    left, right = np.floor(in_x).astype(np.int64).clip(0, W - 1), np.ceil(in_x).astype(np.int64).clip(0, W - 1)
    ly = torch.from_numpy((in_y - np.floor(in_y)).astype(np.float64)).to(feat.dtype)[None, :, :, None]
    lx = torch.from_numpy((in_x - np.floor(in_x)).astype(np.float64)).to(feat.dtype)[None, :, None, :]
    f = feat[0]
    tt, bb, ll, rr = (torch.from_numpy(a) for a in (top, bot, left, right))
    tl = f[:, tt[:, :, None], ll[:, None, :]]                                # [C,R,P,P]
    tr = f[:, tt[:, :, None], rr[:, None, :]]
    bl = f[:, bb[:, :, None], ll[:, None, :]]
    br = f[:, bb[:, :, None], rr[:, None, :]]
    t = tl + (tr - tl) * lx
    b = bl + (br - bl) * lx
    out = t + (b - t) * ly
    mask = torch.from_numpy((vy[:, :, None] & vx[:, None, :]).astype(np.float64)).to(feat.dtype)[None]
    return (out * mask).permute(1, 0, 2, 3)                                  # [R,C,P,P]


class TrainRef(DenseRef):
    """Reference TRAIN step (network.py:279-321 losses, train_val.py:128-145 solver) in torch float64 autograd,
    given the sampled rois / targets as constants (they carry no gradient in the reference either: py_func,
    tf.stop_gradient, network.py:153)."""

    def __init__(self, variables, num_layers, num_classes, anchor_scales, anchor_ratios, trainable, dtype=torch.float64):
        DenseRef.__init__(self, variables, num_layers, num_classes, anchor_scales, anchor_ratios, dtype)
        self.trainable = trainable

    def w(self, name):
        if name not in self._cache:
            a = self.v[name]
            t = _t(a, self.dtype)
            if name.endswith("/weights") or name.endswith("/biases"):
                scope = name.rsplit("/", 1)[0]
                if self.trainable(scope) and "BatchNorm" not in name:
                    t.requires_grad_(True)
            self._cache[name] = t
        t = self._cache[name]
        return t.permute(3, 2, 0, 1) if t.ndim == 4 else t

    def train_tail(self, feat, rois):
        """RoI pooling + per-RoI tail as differentiable torch ops (resnet: direct 7x7 crop, resnet_v1.py:55-76,115-125)."""
        pool5 = crop_and_resize_torch(feat, rois, 16.0, 7)
        return self.run_blocks(pool5, self.blocks[3:]).mean(dim=(2, 3))

    def losses(self, image_nhwc, rois, at, pt, sigma_rpn=3.0):
        feat = self.head(image_nhwc)
        s, A = self.scope, self.A
        r = F.relu(F.conv2d(feat, self.w(s + "/rpn_conv/3x3/weights"), self.w(s + "/rpn_conv/3x3/biases"), padding=1))
        score = F.conv2d(r, self.w(s + "/rpn_cls_score/weights"), self.w(s + "/rpn_cls_score/biases"))[0]     # [2A,H,W]
        bbox = F.conv2d(r, self.w(s + "/rpn_bbox_pred/weights"), self.w(s + "/rpn_bbox_pred/biases"))
        pair = torch.stack([score[:A].reshape(-1), score[A:].reshape(-1)], dim=1)                                # (a,h,w) order
        lab = torch.from_numpy(np.asarray(at["rpn_labels"]).ravel()).long()
        sel = lab >= 0
        rpn_ce = F.cross_entropy(pair[sel], lab[sel])
        sl1 = lambda pred, tg, iw, ow, sg: (ow * torch.where((iw * (pred - tg)).abs() < 1.0 / sg ** 2,
                                                               (iw * (pred - tg)) ** 2 * (sg ** 2 / 2.0),
                                                               (iw * (pred - tg)).abs() - 0.5 / sg ** 2)).sum()
        d = lambda k, src: _t(np.asarray(src[k]), self.dtype)
        rpn_box = sl1(bbox.permute(0, 2, 3, 1), d("rpn_bbox_targets", at), d("rpn_bbox_inside_weights", at),
                      d("rpn_bbox_outside_weights", at), sigma_rpn)
        fc7 = self.train_tail(feat, rois)
        cls_score = fc7 @ self.w(s + "/cls_score/weights") + self.w(s + "/cls_score/biases")
        bbox_pred = fc7 @ self.w(s + "/bbox_pred/weights") + self.w(s + "/bbox_pred/biases")
        ce = F.cross_entropy(cls_score, torch.from_numpy(np.asarray(pt["labels"]).ravel()).long())
        box = sl1(bbox_pred, d("bbox_targets", pt), d("bbox_inside_weights", pt), d("bbox_outside_weights", pt), 1.0) / bbox_pred.shape[0]
        return dict(rpn_cross_entropy=rpn_ce, rpn_loss_box=rpn_box, cross_entropy=ce, loss_box=box)


class VGG16Ref(DenseRef):
    """lib/nets/vgg16.py:26-60 (TEST): conv+bias+ReLU x13, 2x2/2 SAME max pools, fc6/fc7 on the
    NHWC-flattened 7x7x512 crop."""
    max_pool_crop = True
    CFG = [("conv1", 2), ("conv2", 2), ("conv3", 3), ("conv4", 3), ("conv5", 3)]

    def __init__(self, variables, num_classes=21, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2), dtype=torch.float64):
        DenseRef.__init__(self, variables, 50, num_classes, anchor_scales, anchor_ratios, dtype)
        self.scope = "vgg_16"

    def head(self, image_nhwc):
        x = _t(image_nhwc, self.dtype).permute(0, 3, 1, 2)
        for bi, (name, reps) in enumerate(self.CFG):
            for r in range(1, reps + 1):
                sc = "%s/%s/%s_%d" % (self.scope, name, name, r)
                x = F.relu(F.conv2d(x, self.w(sc + "/weights"), self.w(sc + "/biases"), padding=1))
            if bi < 4:
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)          # 'SAME': pad bottom/right, ignored
        return x

    def tail(self, pool5_nhwc):
        x = _t(pool5_nhwc, self.dtype).reshape(pool5_nhwc.shape[0], -1)        # slim.flatten: (h, w, c)
        w6 = _t(np.asarray(self.v[self.scope + "/fc6/weights"]).reshape(-1, 4096), self.dtype)
        x = F.relu(x @ w6 + self.w(self.scope + "/fc6/biases"))
        return F.relu(x @ self.w(self.scope + "/fc7/weights") + self.w(self.scope + "/fc7/biases"))


class MobileNetRef(DenseRef):
    """lib/nets/mobilenet_v1.py:63-79,114-172,214-250 (TEST), depth multiplier 1.0."""
    max_pool_crop = True
    SEP = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512), (1, 512), (1, 512), (1, 512),
           (1, 1024), (1, 1024)]

    def __init__(self, variables, num_classes=21, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2), dtype=torch.float64):
        DenseRef.__init__(self, variables, 50, num_classes, anchor_scales, anchor_ratios, dtype)
        self.scope = "MobilenetV1"

    def sep(self, x, i, stride):
        dw = "%s/Conv2d_%d_depthwise" % (self.scope, i)
        w = _t(np.transpose(self.v[dw + "/depthwise_weights"], (2, 3, 0, 1)), self.dtype)      # [C,1,3,3]
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), w, stride=stride, groups=x.shape[1])
        x = self.bn(x, dw, eps=1e-3).clamp(0, 6)
        pw = "%s/Conv2d_%d_pointwise" % (self.scope, i)
        return self.bn(F.conv2d(x, self.w(pw + "/weights")), pw, eps=1e-3).clamp(0, 6)

    def head(self, image_nhwc):
        x = _t(image_nhwc, self.dtype).permute(0, 3, 1, 2)
        x = self.bn(self.conv_same(x, self.scope + "/Conv2d_0", 3, 2), self.scope + "/Conv2d_0", eps=1e-3).clamp(0, 6)
        for i in range(1, 12):
            x = self.sep(x, i, self.SEP[i - 1][0])
        return x

    def tail(self, pool5_nhwc):
        x = _t(pool5_nhwc, self.dtype).permute(0, 3, 1, 2)
        for i in (12, 13):
            x = self.sep(x, i, self.SEP[i - 1][0])
        return x.mean(dim=(2, 3))


class VGG16TrainRef(TrainRef, VGG16Ref):
    """VGG16 TRAIN graph (vgg16.py:26-60, network.py:141-157): 14x14 crop + 2x2 max, fc6 / fc7 with dropout; the two dropout
    masks (already divided by keep_prob) are given as constants, like the sampled rois / targets."""

    def __init__(self, variables, num_classes, anchor_scales, anchor_ratios, trainable, masks, dtype=torch.float64):
        VGG16Ref.__init__(self, variables, num_classes, anchor_scales, anchor_ratios, dtype)
        self.trainable = trainable
        self.masks = [_t(m, dtype) for m in masks]

    def train_tail(self, feat, rois):
        s = self.scope
        pool5 = F.max_pool2d(crop_and_resize_torch(feat, rois, 16.0, 14), 2, 2)                   # [R,C,7,7]
        x = pool5.permute(0, 2, 3, 1).reshape(pool5.shape[0], -1)                                  # slim.flatten: (h, w, c)
        x = F.relu(x @ self.w(s + "/fc6/weights") + self.w(s + "/fc6/biases")) * self.masks[0]
        return F.relu(x @ self.w(s + "/fc7/weights") + self.w(s + "/fc7/biases")) * self.masks[1]


class MobileNetTrainRef(TrainRef, MobileNetRef):
    """MobileNet-v1 TRAIN graph (mobilenet_v1.py:214-250): frozen batch norm everywhere, layers >= FIXED_LAYERS train their
    depthwise and pointwise filters."""

    def __init__(self, variables, num_classes, anchor_scales, anchor_ratios, trainable, dtype=torch.float64):
        MobileNetRef.__init__(self, variables, num_classes, anchor_scales, anchor_ratios, dtype)
        self.trainable = trainable

    def w_dw(self, scope):
        name = scope + "/depthwise_weights"
        if name not in self._cache:
            t = _t(self.v[name], self.dtype)                                                       # [3,3,C,1]
            if self.trainable(scope):
                t.requires_grad_(True)
            self._cache[name] = t
        return self._cache[name].permute(2, 3, 0, 1)                                               # [C,1,3,3]

    def sep(self, x, i, stride):
        dw = "%s/Conv2d_%d_depthwise" % (self.scope, i)
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), self.w_dw(dw), stride=stride, groups=x.shape[1])
        x = self.bn(x, dw, eps=1e-3).clamp(0, 6)
        pw = "%s/Conv2d_%d_pointwise" % (self.scope, i)
        return self.bn(F.conv2d(x, self.w(pw + "/weights")), pw, eps=1e-3).clamp(0, 6)

    def train_tail(self, feat, rois):
        x = F.max_pool2d(crop_and_resize_torch(feat, rois, 16.0, 14), 2, 2)
        for i in (12, 13):
            x = self.sep(x, i, self.SEP[i - 1][0])
        return x.mean(dim=(2, 3))
