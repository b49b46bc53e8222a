"""crop_and_resize at the bench step's shapes (8 images, 300 RoIs each, 38 x 63 feature maps): the workgroup-per-(roi, slab) form of round 5
against the per-(roi, output row, slab) form of rounds 2-4 (frcnn_detect_set_tuning key 5), slab counts, bit equality, algorithmic GB/s.

    python scratch/crop_bench.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tf-faster-rcnn_amd")]
import numpy as np, torch
import frcnn_hip
from frcnn_hip import ops
dev = torch.device("cuda:0")
L = frcnn_hip.lib()
rng = np.random.RandomState(0)
N, H, W, R = 8, 38, 63, 2400
# proposal-like boxes in image coordinates (600 x 1000): sizes log-uniform 32 ... 600
w = np.exp(rng.uniform(np.log(32), np.log(900), R)); h = np.exp(rng.uniform(np.log(32), np.log(560), R))
x1 = rng.uniform(0, np.maximum(1000 - w, 1)); y1 = rng.uniform(0, np.maximum(600 - h, 1))
rois = np.stack([np.repeat(np.arange(N), R // N), x1, y1, np.minimum(x1 + w, 999), np.minimum(y1 + h, 599)], 1).astype(np.float32)
rois_d = torch.from_numpy(rois).to(dev)
print("%-22s %-28s %9s %9s %8s" % ("shape", "form", "med_us", "min_us", "GB/s"))
for C, pool, mx in ((512, 7, False), (2048, 7, False), (1024, 7, False), (512, 7, True)):
    feat = torch.randn(N, H, W, C, device=dev)
    outs, res = {}, {}
    forms = [("per (roi, row, slab)", 1, -1), ("per (roi, slab)", 0, -1), ("per (roi, slab), 4 slabs", 0, 4), ("per (roi, slab), 16 slabs", 0, 16), ("per (roi, slab), 1 slab", 0, 1)]
    for name, form, slabs in forms:
        L.frcnn_detect_set_tuning(5, form); L.frcnn_detect_set_tuning(4, slabs)
        out = torch.full((R, pool, pool, C), float("nan"), device=dev)
        ops.crop_and_resize(feat, rois_d, 16.0, pool, max_pool=mx, out=out); torch.cuda.synchronize()
        outs[name] = out
    base = outs[forms[0][0]]
    times = {f[0]: [] for f in forms}
    for rep in range(6):
        for name, form, slabs in forms:
            L.frcnn_detect_set_tuning(5, form); L.frcnn_detect_set_tuning(4, slabs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.crop_and_resize(feat, rois_d, 16.0, pool, max_pool=mx, out=outs[name])
            e1.record(); torch.cuda.synchronize()
            if rep: times[name].append(e0.elapsed_time(e1) * 1000 / 5)
    nbytes = 4.0 * (feat.numel() + base.numel())
    for name, _, _ in forms:
        med = float(np.median(times[name]))
        print("C %4d pool %d max %d    %-28s %9.1f %9.1f %8.0f  %s" % (C, pool, int(mx), name, med, min(times[name]), nbytes / med / 1e3,
              "identical" if torch.equal(outs[name].view(torch.int32), base.view(torch.int32)) else "DIFFERENT"))
L.frcnn_detect_set_tuning(5, 0); L.frcnn_detect_set_tuning(4, -1)
