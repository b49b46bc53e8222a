import sys, os
sys.path[:0] = ["oracle", "tf-faster-rcnn_amd"]
import numpy as np, torch
import frcnn_oracle as ora, synth
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
def au(x, a=256): return (x + a - 1) // a * a
for k, thr, cl in [(2, 0.5, 0), (64, 0.3, 2), (65, 0.3, 2), (200, 0.5, 3), (3000, 0.7, 0)]:
    d = synth.random_dets(k, seed=11, cluster=cl)
    keep, num = ops.nms(torch.from_numpy(d).to(dev), thr)
    torch.cuda.synchronize()
    n = int(num.item()); got = keep[:n].cpu().numpy().tolist(); want = ora.cpu_nms(d, thr)
    ws = ops._ws_cache[(str(dev), "nms", "default")].cpu().numpy()
    off = 0
    def take(nb):
        global off
        o = off; off = au(off + nb); return o
    o_boxes = take(16 * k); o_keys = take(8 * k); o_rank = take(4 * k); o_sb = take(16 * k); o_ss = take(4 * k); o_si = take(4 * k); o_mask = take(8 * k * ((k + 63) // 64))
    cb = (k + 63) // 64
    rank = ws[o_rank:o_rank + 4 * k].view(np.uint32)
    sidx = ws[o_si:o_si + 4 * k].view(np.int32)
    sb = ws[o_sb:o_sb + 16 * k].view(np.float32).reshape(k, 4)
    mask = ws[o_mask:o_mask + 8 * k * cb].view(np.uint64).reshape(k, cb)
    order = ora.order_desc(d[:, 4])
    print("k", k, "keep ok", got == want, n, len(want), "rank perm", sorted(rank.tolist()) == list(range(k)), "order ok", np.array_equal(sidx, order), "boxes ok", np.array_equal(sb, d[order, :4]))
    # mask check
    ds = d[order]
    thr_f = np.float32(thr)
    if float(thr_f) < thr: thr_f = np.nextafter(thr_f, np.float32(np.inf))
    x1, y1, x2, y2 = ds[:, 0], ds[:, 1], ds[:, 2], ds[:, 3]
    area = ((x2 - x1) + np.float32(1)) * ((y2 - y1) + np.float32(1))
    bad = 0
    for i in range(min(k, 400)):
        xx1 = np.maximum(x1[i], x1); yy1 = np.maximum(y1[i], y1); xx2 = np.minimum(x2[i], x2); yy2 = np.minimum(y2[i], y2)
        w = np.maximum(np.float32(0), (xx2 - xx1) + np.float32(1)); h = np.maximum(np.float32(0), (yy2 - yy1) + np.float32(1))
        inter = w * h; ovr = inter / ((area[i] + area) - inter)
        sup = (ovr >= thr_f) & (np.arange(k) > i)
        for c in range(i // 64, cb):
            bits = 0
            for j in range(c * 64, min(k, c * 64 + 64)):
                if sup[j]: bits |= (1 << (j - c * 64))
            if int(mask[i, c]) != bits:
                bad += 1
                if bad < 4: print("  mask mismatch i", i, "c", c, hex(int(mask[i, c])), hex(bits))
    print("  mask bad words:", bad)
