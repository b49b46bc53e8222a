"""Second, independent statements of the THIRD-PARTY arithmetic the oracle restates (TensorFlow r1.2 / tf.contrib.slim, absent
from /root/reference, not installable here: SURVEY.md 8c "parity unpinned").  Nothing can pin those rows against the real
library, so -- like tf.image.non_max_suppression in tests/test_oracle_golden.py -- each restatement is written twice, by
different means, and the two must agree: a transcription slip in one of them fails here instead of passing every test.

  * tf.image.crop_and_resize: oracle/oracle_c.c (C, pointer walk)  vs  a scalar float32 Python transcription of the published
    CropAndResize CPU functor loop (core/kernels/crop_and_resize_op.cc) + the caller's normalisation (nets/network.py:141-157,
    nets/resnet_v1.py:55-76);
  * slim conv2d / conv2d_same / frozen batch_norm / max_pool2d / bottleneck_v1 / resnet_v1_block stride placement / the RPN pair
    softmax: oracle/dense_ref.py (torch, NCHW, F.conv2d)  vs  plain numpy loops in NHWC written from the slim definitions
    (resnet_utils.conv2d_same: explicit pad (k-1)//2 | rest then VALID; subsample = every stride-th pixel; bottleneck shortcut
    rule; stride on the LAST unit of a block; `SAME` pooling pads bottom/right and ignores the padding)."""
import numpy as np
import torch

import frcnn_oracle as ora
from dense_ref import DenseRef, VGG16Ref

f32 = np.float32


# ------------------------------------------------------------------------------------------------ crop_and_resize
def crop_and_resize_scalar(image, boxes, box_ind, crop_h, crop_w, extrapolation_value=0.0):
    """The CropAndResize functor, scalar by scalar in float32.  image [B,H,W,D], boxes [n,4] = (y1,x1,y2,x2) normalised."""
    _, H, W, D = image.shape
    out = np.zeros((boxes.shape[0], crop_h, crop_w, D), dtype=f32)
    for b in range(boxes.shape[0]):
        y1, x1, y2, x2 = (f32(v) for v in boxes[b])
        b_in = int(box_ind[b])
        height_scale = (y2 - y1) * f32(H - 1) / f32(crop_h - 1) if crop_h > 1 else f32(0)
        width_scale = (x2 - x1) * f32(W - 1) / f32(crop_w - 1) if crop_w > 1 else f32(0)
        for y in range(crop_h):
            in_y = y1 * f32(H - 1) + f32(y) * height_scale if crop_h > 1 else f32(0.5) * (y1 + y2) * f32(H - 1)
            if in_y < 0 or in_y > H - 1:
                out[b, y, :, :] = extrapolation_value
                continue
            top, bottom = int(np.floor(in_y)), int(np.ceil(in_y))
            y_lerp = in_y - f32(top)
            for x in range(crop_w):
                in_x = x1 * f32(W - 1) + f32(x) * width_scale if crop_w > 1 else f32(0.5) * (x1 + x2) * f32(W - 1)
                if in_x < 0 or in_x > W - 1:
                    out[b, y, x, :] = extrapolation_value
                    continue
                left, right = int(np.floor(in_x)), int(np.ceil(in_x))
                x_lerp = in_x - f32(left)
                for d in range(D):
                    tl, tr = image[b_in, top, left, d], image[b_in, top, right, d]
                    bl, br = image[b_in, bottom, left, d], image[b_in, bottom, right, d]
                    t = tl + (tr - tl) * x_lerp
                    bo = bl + (br - bl) * x_lerp
                    out[b, y, x, d] = t + (bo - t) * y_lerp
    return out


def crop_pool_layer_scalar(bottom, rois, feat_stride, pool, max_pool):
    """nets/network.py:141-157 (max_pool: 2*pool crop then 2x2/2 max) / nets/resnet_v1.py:55-76 around the functor above."""
    H, W = bottom.shape[1], bottom.shape[2]
    height = (f32(H) - f32(1)) * f32(feat_stride)                   # (tf.to_float(bottom_shape[1]) - 1.) * feat_stride
    width = (f32(W) - f32(1)) * f32(feat_stride)
    x1, y1 = rois[:, 1] / width, rois[:, 2] / height
    x2, y2 = rois[:, 3] / width, rois[:, 4] / height
    boxes = np.stack([y1, x1, y2, x2], axis=1).astype(f32)          # tf.concat([y1, x1, y2, x2], 1)
    P = pool * 2 if max_pool else pool
    crops = crop_and_resize_scalar(bottom, boxes, rois[:, 0].astype(np.int32), P, P)
    if max_pool:                                                     # slim.max_pool2d(crops, [2, 2], padding='SAME'), stride 2
        crops = crops.reshape(crops.shape[0], pool, 2, pool, 2, crops.shape[3]).max(axis=(2, 4))
    return crops


def test_crop_and_resize_two_statements_agree():
    rng = np.random.RandomState(3)
    H, W, C = 9, 13, 5
    feat = rng.randn(1, H, W, C).astype(f32)
    x1 = rng.rand(40) * (W * 16 - 40)
    y1 = rng.rand(40) * (H * 16 - 40)
    rois = np.stack([np.zeros(40), x1, y1, x1 + 8 + rng.rand(40) * 150, y1 + 8 + rng.rand(40) * 100], axis=1).astype(f32)
    rois[:, 3] = np.minimum(rois[:, 3], W * 16 - 1)                  # clipped to the IMAGE (bbox_transform.py:74-80) -> beyond the
    rois[:, 4] = np.minimum(rois[:, 4], H * 16 - 1)                  # last feature row/column: zeros, not clamps
    rois[0] = [0, 0, 0, W * 16 - 1, H * 16 - 1]                      # the whole image
    rois[1] = [0, 16, 16, 16, 16]                                    # a single point exactly on a feature pixel
    rois[2] = [0, 30.5, 40.25, 30.5, 90.0]                           # zero width
    for pool, mp in ((7, False), (7, True), (3, False)):
        want = ora.crop_and_resize(feat[0], rois, 16.0, pool, max_pool=mp)
        got = crop_pool_layer_scalar(feat, rois, 16.0, pool, mp)
        assert got.shape == want.shape and np.array_equal(got, want), (pool, mp, float(np.abs(got - want).max()))
    assert np.any(ora.crop_and_resize(feat[0], rois, 16.0, 7)[0, -1] == 0)       # the border-zero quirk is exercised


# ------------------------------------------------------------------------------------------------ slim layers
def conv2d_loops(x, w_hwio, stride, pad):
    """x [N,H,W,Cin] float64, w [KH,KW,Cin,Cout]; pad = (top, bottom, left, right) zeros, then VALID."""
    N, H, W, Cin = x.shape
    KH, KW, _, Cout = w_hwio.shape
    xp = np.zeros((N, H + pad[0] + pad[1], W + pad[2] + pad[3], Cin))
    xp[:, pad[0]:pad[0] + H, pad[2]:pad[2] + W, :] = x
    OH, OW = (xp.shape[1] - KH) // stride + 1, (xp.shape[2] - KW) // stride + 1
    y = np.zeros((N, OH, OW, Cout))
    for oh in range(OH):
        for ow in range(OW):
            patch = xp[:, oh * stride:oh * stride + KH, ow * stride:ow * stride + KW, :]          # [N,KH,KW,Cin]
            y[:, oh, ow, :] = np.tensordot(patch, w_hwio, axes=([1, 2, 3], [0, 1, 2]))
    return y


def conv2d_same_loops(x, w, stride):
    """resnet_utils.conv2d_same: stride 1 -> slim conv2d 'SAME'; else explicit pad pad_beg = (k-1)//2, pad_end = (k-1) - pad_beg."""
    k = w.shape[0]
    total = k - 1
    beg = total // 2
    return conv2d_loops(x, w, stride, (beg, total - beg, beg, total - beg))


def bn_loops(x, v, scope, eps):
    g, b = v[scope + "/BatchNorm/gamma"].astype(np.float64), v[scope + "/BatchNorm/beta"].astype(np.float64)
    m, var = v[scope + "/BatchNorm/moving_mean"].astype(np.float64), v[scope + "/BatchNorm/moving_variance"].astype(np.float64)
    return (x - m) / np.sqrt(var + eps) * g + b


def max_pool_loops(x, k, stride, pad):
    """max over the window; padded cells never win (TF pads max-pool with -inf semantics)."""
    N, H, W, C = x.shape
    xp = np.full((N, H + pad[0] + pad[1], W + pad[2] + pad[3], C), -np.inf)
    xp[:, pad[0]:pad[0] + H, pad[2]:pad[2] + W, :] = x
    OH, OW = (xp.shape[1] - k) // stride + 1, (xp.shape[2] - k) // stride + 1
    y = np.zeros((N, OH, OW, C))
    for oh in range(OH):
        for ow in range(OW):
            y[:, oh, ow, :] = xp[:, oh * stride:oh * stride + k, ow * stride:ow * stride + k, :].max(axis=(1, 2))
    return y


def bottleneck_loops(x, v, p, base, stride):
    """slim resnet_v1.bottleneck: shortcut = subsample(inputs, stride) if depth_in == depth else conv 1x1 stride (BN, no act);
    residual = conv1 1x1/1 -> conv2 3x3 conv2d_same(stride) -> conv3 1x1/1 (no act); relu(shortcut + residual)."""
    depth = base * 4
    relu = lambda a: np.maximum(a, 0)
    W = lambda s: v[p + s + "/weights"].astype(np.float64)
    if x.shape[-1] == depth:
        sc = x if stride == 1 else x[:, ::stride, ::stride, :]
    else:
        sc = bn_loops(conv2d_loops(x, W("/shortcut"), stride, (0, 0, 0, 0)), v, p + "/shortcut", 1e-5)
    r = relu(bn_loops(conv2d_loops(x, W("/conv1"), 1, (0, 0, 0, 0)), v, p + "/conv1", 1e-5))
    r = relu(bn_loops(conv2d_same_loops(r, W("/conv2"), stride), v, p + "/conv2", 1e-5))
    r = bn_loops(conv2d_loops(r, W("/conv3"), 1, (0, 0, 0, 0)), v, p + "/conv3", 1e-5)
    return relu(sc + r)


def _tiny_resnet_variables(rng, scope, blocks, cin0=8):
    v = {}

    def conv_bn(s, k, ci, co):
        v[s + "/weights"] = (rng.randn(k, k, ci, co) * np.sqrt(2.0 / (k * k * ci))).astype(f32)
        v[s + "/BatchNorm/gamma"] = rng.uniform(0.5, 1.5, co).astype(f32)
        v[s + "/BatchNorm/beta"] = (rng.randn(co) * 0.1).astype(f32)
        v[s + "/BatchNorm/moving_mean"] = (rng.randn(co) * 0.1).astype(f32)
        v[s + "/BatchNorm/moving_variance"] = rng.uniform(0.5, 1.5, co).astype(f32)
    conv_bn(scope + "/conv1", 7, 3, cin0)
    cin = cin0
    for name, base, n, _ in blocks:
        for u in range(1, n + 1):
            p = "%s/%s/unit_%d/bottleneck_v1" % (scope, name, u)
            if cin != base * 4:
                conv_bn(p + "/shortcut", 1, cin, base * 4)
            conv_bn(p + "/conv1", 1, cin, base)
            conv_bn(p + "/conv2", 3, base, base)
            conv_bn(p + "/conv3", 1, base, base * 4)
            cin = base * 4
    return v, cin


def test_resnet_head_two_statements_agree():
    """conv1 (conv2d_same 7x7/2) + pad-1 3x3/2 VALID pool + two blocks with the stride on the LAST unit, odd sizes."""
    rng = np.random.RandomState(5)
    ref = DenseRef({}, 50, 3, (8,), (1,))
    ref.blocks = [("block1", 2, 2, 2), ("block2", 3, 2, 2), ("block3", 4, 2, 1), ("block4", 4, 1, 1)]
    v, cin = _tiny_resnet_variables(rng, ref.scope, ref.blocks)
    ref.v = v
    image = rng.randn(1, 37, 45, 3).astype(f32)
    with torch.no_grad():
        want = ref.head(image).permute(0, 2, 3, 1).numpy()
    x = image.astype(np.float64)
    x = np.maximum(bn_loops(conv2d_same_loops(x, v[ref.scope + "/conv1/weights"].astype(np.float64), 2), v, ref.scope + "/conv1", 1e-5), 0)
    assert x.shape[1:3] == (19, 23)                                   # (n - 1) // 2 + 1
    x = max_pool_loops(x, 3, 2, (1, 1, 1, 1))                          # resnet_v1.py:83-84: pad [1,1] then 3x3/2 VALID
    assert x.shape[1:3] == (10, 12)
    for name, base, n, stride in ref.blocks[:3]:
        for u in range(1, n + 1):
            x = bottleneck_loops(x, v, "%s/%s/unit_%d/bottleneck_v1" % (ref.scope, name, u), base, stride if u == n else 1)
    assert x.shape == want.shape == (1, 3, 3, 16)                      # 10x12 -> 5x6 -> 3x3 (block3 keeps stride 1)
    assert np.abs(x - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
    # block4 per RoI + spatial mean (resnet_v1.py:115-125)
    pool5 = rng.randn(4, 7, 7, 16).astype(f32)
    with torch.no_grad():
        fc7 = ref.tail(pool5).numpy()
    y = bottleneck_loops(pool5.astype(np.float64), v, ref.scope + "/block4/unit_1/bottleneck_v1", 4, 1).mean(axis=(1, 2))
    assert np.abs(y - fc7).max() <= 1e-11 * max(1.0, np.abs(fc7).max())


def test_rpn_and_heads_two_statements_agree():
    """RPN 3x3 SAME + bias + ReLU, the two 1x1 heads, the pair softmax over channels (a, A+a) (network.py:68-86, 323-337), the
    fc heads and the test-time de-normalisation (network.py:361-378, 428-432)."""
    rng = np.random.RandomState(7)
    A, C, Cf = 3, 4, 6
    ref = DenseRef({}, 50, C, (8,), (0.5, 1, 2))
    s = ref.scope
    v = {s + "/rpn_conv/3x3/weights": (rng.randn(3, 3, Cf, 5) * 0.2).astype(f32), s + "/rpn_conv/3x3/biases": rng.randn(5).astype(f32) * 0.1,
         s + "/rpn_cls_score/weights": (rng.randn(1, 1, 5, 2 * A)).astype(f32), s + "/rpn_cls_score/biases": rng.randn(2 * A).astype(f32) * 0.1,
         s + "/rpn_bbox_pred/weights": (rng.randn(1, 1, 5, 4 * A) * 0.1).astype(f32), s + "/rpn_bbox_pred/biases": rng.randn(4 * A).astype(f32) * 0.1,
         s + "/cls_score/weights": rng.randn(8, C).astype(f32), s + "/cls_score/biases": rng.randn(C).astype(f32),
         s + "/bbox_pred/weights": (rng.randn(8, 4 * C) * 0.1).astype(f32), s + "/bbox_pred/biases": rng.randn(4 * C).astype(f32) * 0.1}
    ref.v = v
    feat = rng.randn(1, 5, 7, Cf)
    with torch.no_grad():
        score, prob, bbox = ref.rpn(torch.from_numpy(feat).permute(0, 3, 1, 2))
    d = lambda k: v[k].astype(np.float64)
    r = np.maximum(conv2d_loops(feat, d(s + "/rpn_conv/3x3/weights"), 1, (1, 1, 1, 1)) + d(s + "/rpn_conv/3x3/biases"), 0)
    sc = conv2d_loops(r, d(s + "/rpn_cls_score/weights"), 1, (0, 0, 0, 0)) + d(s + "/rpn_cls_score/biases")
    bb = conv2d_loops(r, d(s + "/rpn_bbox_pred/weights"), 1, (0, 0, 0, 0)) + d(s + "/rpn_bbox_pred/biases")
    # _reshape_layer(…, 2): NHWC -> NCHW -> [1, 2, A*H, W] -> softmax over dim 1 -> back: channel a pairs with channel A + a
    pr = np.zeros_like(sc)
    for a in range(A):
        e0, e1 = np.exp(sc[..., a]), np.exp(sc[..., A + a])
        pr[..., a], pr[..., A + a] = e0 / (e0 + e1), e1 / (e0 + e1)
    assert np.abs(sc - score).max() < 1e-12 and np.abs(bb - bbox).max() < 1e-12 and np.abs(pr - prob).max() < 1e-12
    fc7 = rng.randn(6, 8)
    cs, cp, bp = ref.classify(torch.from_numpy(fc7))
    logits = fc7 @ d(s + "/cls_score/weights") + d(s + "/cls_score/biases")
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    bpred = (fc7 @ d(s + "/bbox_pred/weights") + d(s + "/bbox_pred/biases")) * np.tile((0.1, 0.1, 0.2, 0.2), C) + np.tile((0.0,) * 4, C)
    assert np.abs(cs - logits).max() < 1e-12 and np.abs(cp - e / e.sum(axis=1, keepdims=True)).max() < 1e-12
    assert np.abs(bp - bpred).max() < 1e-12


def test_vgg_same_pooling_two_statements_agree():
    """slim.max_pool2d(k=2, s=2, 'SAME') on odd sizes: out = ceil(n / 2), padding at the bottom / right, ignored (vgg16.py:26-46)."""
    rng = np.random.RandomState(9)
    x = rng.randn(1, 7, 9, 3)
    want = torch.nn.functional.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1).numpy()   # VGG16Ref.head
    got = max_pool_loops(x, 2, 2, (0, 7 % 2, 0, 9 % 2))
    assert got.shape == want.shape == (1, 4, 5, 3) and np.array_equal(got, want)
    assert VGG16Ref.max_pool_crop is True
