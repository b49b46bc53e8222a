"""Isolated timing of the filter-gradient kernels on the ResNet-152 / 600x1000 training shapes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd")); sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd", "lib"))
from frcnn_hip import lib, ops
dev = torch.device("cuda:0")
SHAPES = [  # name, N, H, W, Cin, Cout, k, stride, pad
    ("b3 conv1 1x1 1024->256", 1, 38, 63, 1024, 256, 1, 1, (0, 0, 0, 0)),
    ("b3 conv2 3x3 256->256", 1, 38, 63, 256, 256, 3, 1, (1, 1, 1, 1)),
    ("b3 conv3 1x1 256->1024", 1, 38, 63, 256, 1024, 1, 1, (0, 0, 0, 0)),
    ("b2 conv1 1x1 512->128", 1, 75, 125, 512, 128, 1, 1, (0, 0, 0, 0)),
    ("b2 conv2 3x3 128->128", 1, 75, 125, 128, 128, 3, 1, (1, 1, 1, 1)),
    ("b2 conv3 1x1 128->512", 1, 75, 125, 128, 512, 1, 1, (0, 0, 0, 0)),
    ("rpn 3x3 1024->512", 1, 38, 63, 1024, 512, 3, 1, (1, 1, 1, 1)),
    ("tail conv1 1x1 2048->512", 256, 7, 7, 2048, 512, 1, 1, (0, 0, 0, 0)),
    ("tail conv2 3x3 512->512", 256, 7, 7, 512, 512, 3, 1, (1, 1, 1, 1)),
    ("tail conv3 1x1 512->2048", 256, 7, 7, 512, 2048, 1, 1, (0, 0, 0, 0)),
]
plans = [("f32", False, (0, 0)), ("h2 auto", True, (0, 0)), ("h2 t64", True, (64, 0)), ("h2 t128", True, (128, 0)), ("h2 t64 w1024", True, (64, 1024)),
         ("h2 t128 w256", True, (128, 256))]
for name, N, H, W, Cin, Cout, k, st, pad in SHAPES:
    OH = (H + pad[0] + pad[1] - k) // st + 1; OW = (W + pad[2] + pad[3] - k) // st + 1
    x = torch.relu(torch.randn(N, H, W, Cin, device=dev)); gy = torch.randn(N, OH, OW, Cout, device=dev)
    out = torch.empty(Cout, k, k, Cin, device=dev)
    fl = 2.0 * N * OH * OW * Cout * k * k * Cin
    row = []
    for pname, h2, plan in plans:
        setter = lib().frcnn_conv2d_wgrad_h2_set_plan if h2 else lib().frcnn_conv2d_wgrad_set_plan
        setter(*plan)
        for _ in range(3): ops.conv2d_wgrad(gy, x, k, k, st, pad, out, h2=h2)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d_wgrad(gy, x, k, k, st, pad, out, h2=h2)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        setter(0, 0)
        row.append("%s %.0f us %.0f TF" % (pname, us, fl / us / 1e6))
    print("%-28s %s" % (name, " | ".join(row)))
