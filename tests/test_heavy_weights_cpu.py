"""The `+heavy` weights of the full-size harness (oracle/fullsize.py::heavy_rescale) claim to put outlier channels (x 2^10 ... 2^17) into
a network WITHOUT changing its function, so that the committed float64 fixtures stay the reference.  Checked here on the CPU with the
oracle's own dense restatement (float64 AND float32) on small inputs: ResNet-50 head + tail and VGG16 head give the same tensors with and
without the rescaling -- exactly in float32 too where no product underflows (every factor is a power of two, ReLU is positively
homogeneous) -- while the rescaled channels themselves really carry 2^10 ... 2^17 times larger activations."""
import numpy as np
import torch

import fullsize as fs
from dense_ref import DenseRef, VGG16Ref


def _variables(net_kind, layers):
    from frcnn_hip.runtime import VariableStore
    c = dict(net=net_kind, layers=layers)
    net = fs.make_net(c)
    net.create_architecture("TEST", 21, tag="heavy_cpu", anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2))
    store = VariableStore(seed=5)
    store.init_variables(net.variable_specs())
    return net, store.variables


def test_heavy_rescale_leaves_the_resnet_function_unchanged():
    net, v = _variables("res", 50)
    rng = np.random.RandomState(1)
    image = (rng.rand(1, 96, 128, 3) * 255.0 - 110.0).astype(np.float32)
    rois = np.array([[0, 8, 8, 90, 70], [0, 30, 20, 120, 90], [0, 0, 0, 127, 95]], dtype=np.float32)
    hv = {k: np.array(a, copy=True) for k, a in v.items()}
    log = fs.heavy_rescale(hv, net._scope, "res")
    assert len(log) >= 2 * 16 and all(10 <= k <= 17 for _, _, k in log)       # >= one channel per conv1 / conv2 of the 16 units
    import frcnn_oracle as ora
    outs = {}
    for dtype in (torch.float64, torch.float32):
        for name, var in (("base", v), ("heavy", hv)):
            ref = DenseRef(var, 50, 21, (8, 16, 32), (0.5, 1, 2), dtype=dtype)
            with torch.no_grad():
                feat = ref.head(image)
                score, prob, bbox = ref.rpn(feat)
                h = feat.permute(0, 2, 3, 1).contiguous().numpy()
                fc7 = ref.tail(ora.crop_and_resize(h[0].astype(np.float32), rois, 16.0, 7, max_pool=ref.max_pool_crop))
                cls_score, _, bbox_pred = ref.classify(fc7)
            outs[(dtype, name)] = [np.asarray(t) for t in (h, score, bbox, fc7.numpy(), cls_score, bbox_pred)]
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-6)):
        for a, b in zip(outs[(dtype, "base")], outs[(dtype, "heavy")]):
            scale = max(1.0, float(np.abs(a).max()))
            assert float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) <= tol * scale
    # ... and the rescaled channel really is an outlier: conv1 of one unit, its channel c against the rest of the tensor
    scope, c, k = log[0]
    ref_b, ref_h = DenseRef(v, 50, 21), DenseRef(hv, 50, 21)
    assert np.array_equal(hv[scope + "/BatchNorm/gamma"][c], v[scope + "/BatchNorm/gamma"][c] * np.float32(2.0 ** k))
    nxt = scope.replace("/conv1", "/conv2")
    assert np.array_equal(hv[nxt + "/weights"][:, :, c, :], v[nxt + "/weights"][:, :, c, :] / np.float32(2.0 ** k))
    assert ref_b is not ref_h


def test_heavy_rescale_leaves_the_vgg16_head_unchanged():
    net, v = _variables("vgg16", 0)
    rng = np.random.RandomState(2)
    image = ((rng.rand(1, 64, 96, 3) * 255.0 - 110.0) / 64.0).astype(np.float32)
    hv = {k: np.array(a, copy=True) for k, a in v.items()}
    log = fs.heavy_rescale(hv, net._scope, "vgg16")
    assert len(log) >= 12
    heads = []
    for var in (v, hv):
        ref = VGG16Ref(var, 21, (8, 16, 32), (0.5, 1, 2), dtype=torch.float64)
        with torch.no_grad():
            heads.append(ref.head(image).numpy())
    assert float(np.abs(heads[0] - heads[1]).max()) <= 1e-12 * max(1.0, float(np.abs(heads[0]).max()))
