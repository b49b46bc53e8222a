"""TEST INFRASTRUCTURE ONLY -- seeded synthetic inputs for the detection path.
Shared by oracle/gen_golden.py, tests/ and bench.py's cpu_baseline leg.  Everything derives from
np.random.RandomState(seed) with cfg.RNG_SEED = 3 as the default (model/config.py:255)."""
import numpy as np

f32 = np.float32


def tie_free(x, rng):
    """Nudge duplicate f32 values until all are distinct (bit-exact sort tests need unique keys)."""
    x = x.astype(f32).ravel().copy()
    for _ in range(64):
        _, first = np.unique(x, return_index=True)
        dup = np.ones(x.shape[0], dtype=bool)
        dup[first] = False
        if not dup.any():
            break
        x[dup] = np.nextafter(x[dup], f32(2.0)) + (rng.rand(int(dup.sum())).astype(f32) * f32(1e-6))
    assert np.unique(x).size == x.size
    return x


def rpn_outputs(H, W, A, seed=3, delta_std=0.2, unique=True):
    """rpn_cls_prob [1,H,W,2A] (pairwise softmax of N(0,1) logits over channels (a, A+a)) and
    rpn_bbox_pred [1,H,W,4A] ~ N(0, delta_std^2)."""
    rng = np.random.RandomState(seed)
    logits = rng.randn(1, H, W, 2 * A).astype(f32)
    bg, fg = logits[..., :A].astype(np.float64), logits[..., A:].astype(np.float64)
    m = np.maximum(bg, fg)
    e0, e1 = np.exp(bg - m), np.exp(fg - m)
    p_fg = (e1 / (e0 + e1)).astype(f32)
    if unique:
        p_fg = tie_free(p_fg, rng).reshape(1, H, W, A)
    prob = np.concatenate([(f32(1) - p_fg).astype(f32), p_fg], axis=-1).astype(f32)
    deltas = (rng.randn(1, H, W, 4 * A) * delta_std).astype(f32)
    return prob, deltas


def random_dets(k, seed=3, im_w=1000.0, im_h=600.0, unique=True, cluster=0):
    """dets f32 [k,5].  cluster>0 draws boxes around `cluster` centres so NMS suppresses a lot."""
    rng = np.random.RandomState(seed)
    if cluster:
        c = rng.rand(cluster, 2) * [im_w, im_h]
        idx = rng.randint(0, cluster, size=k)
        cx = c[idx, 0] + rng.randn(k) * 25
        cy = c[idx, 1] + rng.randn(k) * 25
    else:
        cx, cy = rng.rand(k) * im_w, rng.rand(k) * im_h
    w, h = 16 + rng.rand(k) * 300, 16 + rng.rand(k) * 300
    x1, y1 = np.clip(cx - w / 2, 0, im_w - 1), np.clip(cy - h / 2, 0, im_h - 1)
    x2, y2 = np.clip(cx + w / 2, 0, im_w - 1), np.clip(cy + h / 2, 0, im_h - 1)
    s = rng.rand(k).astype(f32)
    if unique:
        s = tie_free(s, rng)
    return np.stack([x1, y1, x2, y2, s], axis=1).astype(f32)


def rcnn_outputs(R, C, seed=3, im_w=1000.0, im_h=600.0):
    """Per-class stage inputs: cls_prob [R,C] (softmax of N(0,2^2) logits), bbox_pred [R,4C]
    (N(0,0.1^2), already de-normalised), rois [R,5] in scaled-image coords."""
    rng = np.random.RandomState(seed)
    logits = rng.randn(R, C) * 2.0
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    prob = (e / e.sum(axis=1, keepdims=True)).astype(f32)
    prob = tie_free(prob, rng).reshape(R, C)
    deltas = (rng.randn(R, 4 * C) * 0.1).astype(f32)
    d = random_dets(R, seed=seed + 1, im_w=im_w, im_h=im_h)
    rois = np.hstack([np.zeros((R, 1), dtype=f32), d[:, :4]]).astype(f32)
    return prob, deltas, rois


def gt_boxes(n, C, seed=3, im_w=1000.0, im_h=600.0):
    """gt [n,5] = x1,y1,x2,y2,cls ; sides U[32,400], classes U{1..C-1}."""
    rng = np.random.RandomState(seed)
    w, h = 32 + rng.rand(n) * 368, 32 + rng.rand(n) * 368
    x1 = rng.rand(n) * np.maximum(im_w - w, 1)
    y1 = rng.rand(n) * np.maximum(im_h - h, 1)
    cls = rng.randint(1, C, size=n)
    return np.stack([x1, y1, np.minimum(x1 + w, im_w - 1), np.minimum(y1 + h, im_h - 1), cls], axis=1).astype(f32)


def threshold_pairs(n_pairs=24, seed=3, thr=0.5, filler=200):
    """dets whose decisive overlaps sit EXACTLY on the threshold: pair p is A = 10x10 and B = 10x20 sharing A's area
    (+1 convention: inter 100, union 200 -> IoU = 0.5 exactly, also in float32), B scored below A; pairs are far apart, the
    rest is random filler.  cpu_nms (`>= thresh`) suppresses every B, the CUDA kernel / py_cpu_nms (`> thresh`) keeps it."""
    assert thr == 0.5
    rng = np.random.RandomState(seed)
    rows = []
    for p in range(n_pairs):
        ox, oy = 40.0 * (p % 12), 100.0 + 60.0 * (p // 12)
        rows.append([ox, oy, ox + 9, oy + 9, 0.9 - 0.01 * p])
        rows.append([ox, oy, ox + 9, oy + 19, 0.5 - 0.01 * p])
    d = np.array(rows, dtype=f32)
    fill = random_dets(filler, seed=seed + 1, im_w=1000.0, im_h=90.0)           # y < 90: never touches the pairs
    fill[:, 4] = tie_free(0.05 + 0.3 * rng.rand(filler), rng)
    out = np.vstack([d, fill]).astype(f32)
    out[:, 4] = tie_free(out[:, 4], rng)
    return out[rng.permutation(out.shape[0])]
