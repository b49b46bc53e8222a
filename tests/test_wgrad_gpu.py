"""GPU: frcnn_conv2d_wgrad (csrc/wgrad_tn.hip, f32 matrix pipe) and frcnn_conv2d_wgrad_h2 (csrc/wgrad_h2.hip, two-piece fp16 operands
split in registers) -- the filter gradient of slim.conv2d read straight from dY and X in NHWC -- against
torch's float64 autograd of the same convolution, over the layer kinds of the three backbones' reverse sweeps (pointwise, strided
shortcut, 3x3 SAME, 3x3 stride 2 with conv2d_same's explicit padding, the RoI tail's 7x7 maps, a fully connected layer as a 1x1
convolution), both tile sizes, one slice and many, and a pixel count that is not a multiple of the 32-pixel slab."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, k, stride, pad (top, bottom, left, right)
    ("pointwise", 1, 38, 63, 256, 128, 1, 1, (0, 0, 0, 0)),
    ("shortcut_s2", 1, 38, 63, 128, 256, 1, 2, (0, 0, 0, 0)),
    ("same3x3", 1, 20, 30, 64, 64, 3, 1, (1, 1, 1, 1)),
    ("same3x3_s2", 1, 21, 31, 128, 128, 3, 2, (0, 1, 0, 1)),
    ("roi_tail", 40, 7, 7, 128, 256, 3, 1, (1, 1, 1, 1)),
    ("fc_as_1x1", 96, 1, 1, 1024, 192, 1, 1, (0, 0, 0, 0)),
    ("rpn3x3", 1, 38, 63, 256, 512, 3, 1, (1, 1, 1, 1)),
]


def _reference(x, gy, Cout, k, stride, pad):
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    w = torch.zeros((Cout, x.shape[-1], k, k), dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (pad[2], pad[3], pad[0], pad[1])), w, stride=stride)
    y.backward(torch.from_numpy(gy).double().permute(0, 3, 1, 2))
    return w.grad.permute(0, 2, 3, 1).contiguous().numpy()                 # [Cout, KH, KW, Cin]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("plan", [(0, 0), (64, 1), (128, 4096)], ids=["auto", "t64_one_slice", "t128_many_slices"])
@pytest.mark.parametrize("h2", [False, True], ids=["f32", "h2"])
def test_conv2d_wgrad_vs_float64_autograd(dev, case, plan, h2):
    from frcnn_hip import lib, ops
    _, N, H, W, Cin, Cout, k, stride, pad = case
    OH = (H + pad[0] + pad[1] - k) // stride + 1
    OW = (W + pad[2] + pad[3] - k) // stride + 1
    rng = np.random.RandomState(Cin + Cout + k)
    x = np.maximum(rng.randn(N, H, W, Cin), 0).astype(np.float32)                       # post-ReLU activations
    gy = (rng.randn(N, OH, OW, Cout) * (rng.rand(N, OH, OW, Cout) < 0.4)).astype(np.float32)  # gated gradients
    ref = _reference(x, gy, Cout, k, stride, pad)
    assert ops.conv2d_wgrad_supported(Cin, Cout) and not ops.conv2d_wgrad_supported(Cin, 21) and not ops.conv2d_wgrad_supported(3, Cout)
    out = torch.full((Cout, k, k, Cin), float("nan"), dtype=torch.float32, device=dev)
    setter = lib().frcnn_conv2d_wgrad_h2_set_plan if h2 else lib().frcnn_conv2d_wgrad_set_plan
    setter(*plan)
    try:
        ops.conv2d_wgrad(torch.from_numpy(gy).to(dev), torch.from_numpy(x).to(dev), k, k, stride, pad, out, h2=h2)
        torch.cuda.synchronize()
    finally:
        setter(0, 0)
    got = out.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("wgrad %s %s plan %s: max err / max |dW| = %.2e" % ("h2 " if h2 else "f32", case[0], plan, err))
    assert np.isfinite(got).all() and err <= 2e-6, err


def test_conv2d_wgrad_refuses_what_it_does_not_cover(dev):
    from frcnn_hip import FrcnnHipError, ops
    gy = torch.zeros((1, 8, 8, 21), dtype=torch.float32, device=dev)
    x = torch.zeros((1, 8, 8, 64), dtype=torch.float32, device=dev)
    out = torch.zeros((21, 1, 1, 64), dtype=torch.float32, device=dev)
    with pytest.raises(FrcnnHipError):
        ops.conv2d_wgrad(gy, x, 1, 1, 1, (0, 0, 0, 0), out)
