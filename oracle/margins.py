"""TEST INFRASTRUCTURE ONLY -- decision-margin checks for the data-dependent stages (hard part 1).

The dense part of the graph is float32 arithmetic whose summation order differs between any two implementations (TF/cuDNN,
torch-CPU, the HIP kernels), so scores / boxes agree to ~1e-6, not bit for bit.  Top-N cuts, greedy NMS and the
max_per_image cut are discontinuous in those values: a 1e-7 change can flip a decision whose deciding quantity sits on its
threshold.  "Same result as the reference" for these stages therefore means: the list the device produced is an outcome of
the reference ALGORITHM on the reference's VALUES when every score may move by eps_s and every IoU by eps_iou -- and the
slack it actually needed is reported, so a reader sees 1e-7-level numbers rather than a tolerance taken on trust.

check_greedy_nms verifies, for a kept list in the device's output order (lib/nms/cpu_nms.pyx:17-68 as called from
lib/layer_utils/proposal_layer.py:34-47 and lib/model/test.py:162-180):
  order         consecutive kept scores are non-increasing                                   (slack: score)
  top-N         every kept candidate is inside the pre-NMS top-N                             (slack: score)
  independence  no kept box is suppressed by an earlier kept box: IoU < thr                  (slack: IoU)
  completeness  every candidate that ranks before the end of the list, is inside the top-N and was NOT kept is suppressed
                by a kept box that ranks before it: IoU >= thr                               (slack: IoU and score)
With eps = 0 these four conditions characterise the greedy NMS output exactly (for tie-free scores).
"""
import numpy as np


def iou_matrix(a, b):
    """IoU with the +1 pixel convention of cpu_nms.pyx:57-65, float64.  a [m,4], b [n,4] -> [m,n]."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    w = np.maximum(0.0, np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1.0)
    h = np.maximum(0.0, np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1.0)
    inter = w * h
    aa = (a[:, 2] - a[:, 0] + 1.0) * (a[:, 3] - a[:, 1] + 1.0)
    ab = (b[:, 2] - b[:, 0] + 1.0) * (b[:, 3] - b[:, 1] + 1.0)
    return inter / (aa[:, None] + ab[None, :] - inter)


def match_to_candidates(got_boxes, got_scores, cand_boxes, cand_scores, atol_box, atol_score):
    """Index into the candidates for every output row (-1: no candidate within tolerance).  Exact duplicates among the
    candidates (several anchors clipped to the same box) resolve to the best-scoring unused one."""
    got_boxes = np.asarray(got_boxes, dtype=np.float64).reshape(-1, 4)
    cand_boxes = np.asarray(cand_boxes, dtype=np.float64).reshape(-1, 4)
    cand_scores = np.asarray(cand_scores, dtype=np.float64).ravel()
    used = np.zeros(cand_boxes.shape[0], dtype=bool)
    out = np.full(got_boxes.shape[0], -1, dtype=np.int64)
    for i in range(got_boxes.shape[0]):
        d = np.abs(cand_boxes - got_boxes[i]).max(axis=1)
        ok = (d <= atol_box) & ~used
        if got_scores is not None:
            ok &= np.abs(cand_scores - float(got_scores[i])) <= atol_score
        idx = np.nonzero(ok)[0]
        if idx.size:
            j = idx[np.argmax(cand_scores[idx])]
            out[i] = j
            used[j] = True
    return out


def check_greedy_nms(boxes, scores, kept, thr, eps_s, eps_iou, topn=None, max_keep=None, score_floor=None):
    """boxes [n,4], scores [n]: the candidates in the REFERENCE's arithmetic; kept: candidate indices in the device's output
    order.  max_keep: the list is truncated there (proposal_layer.py:44-45); score_floor: candidates scoring below it are
    outside the checked range (the max_per_image cut of test.py:175-180).  Returns a report dict; report['ok'] is the verdict."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float64).ravel()
    kept = np.asarray(kept, dtype=np.int64).ravel()
    n = scores.shape[0]
    rep = dict(ok=True, n_candidates=int(n), n_kept=int(kept.size), slack_score=0.0, slack_iou=0.0, fragile=0, violations=[])

    def need(kind, what, slack, eps):
        if slack > 0:
            rep["fragile"] += 1
            key = "slack_score" if kind == "s" else "slack_iou"
            rep[key] = max(rep[key], float(slack))
            if slack > eps:
                rep["ok"] = False
                rep["violations"].append((what, float(slack)))

    if kept.size == 0:
        if n and (score_floor is None):
            rep["ok"] = False
            rep["violations"].append(("empty keep list with %d candidates" % n, np.inf))
        return rep
    if (kept < 0).any() or np.unique(kept).size != kept.size:
        rep["ok"] = False
        rep["violations"].append(("unmatched or duplicated rows in the keep list", np.inf))
        return rep
    ks = scores[kept]
    # order
    for t in np.nonzero(ks[1:] > ks[:-1])[0]:
        need("s", "order at position %d" % t, ks[t + 1] - ks[t], eps_s)
    # top-N membership
    cut = -np.inf
    if topn is not None and 0 < topn < n:
        cut = np.partition(scores, n - topn)[n - topn]              # the topn-th largest score
        for t in np.nonzero(ks < cut)[0]:
            need("s", "kept candidate %d below the top-%d cut" % (kept[t], topn), cut - ks[t], eps_s)
    iou_kk = iou_matrix(boxes[kept], boxes[kept])
    # independence: j after i in the list must not be suppressed by i
    iu = np.triu_indices(kept.size, 1)
    over = iou_kk[iu] - thr
    for q in np.nonzero(over >= 0)[0]:
        need("i", "kept %d suppressed by kept %d" % (kept[iu[1][q]], kept[iu[0][q]]), over[q] + 1e-300, eps_iou)
    # completeness
    truncated = max_keep is not None and kept.size >= max_keep
    floor = ks.min() if truncated else -np.inf
    if score_floor is not None:
        floor = max(floor, score_floor)
    is_kept = np.zeros(n, dtype=bool)
    is_kept[kept] = True
    cand = np.nonzero(~is_kept & (scores > floor + eps_s) & (scores >= cut + eps_s))[0]
    if cand.size:
        iou_kc = iou_matrix(boxes[kept], boxes[cand])                 # [kept, cand]
        for q, c in enumerate(cand):
            earlier = ks >= scores[c]                                  # kept boxes that rank before c
            best = iou_kc[earlier, q].max() if earlier.any() else -np.inf
            if best >= thr:
                continue
            # slack needed: either an earlier kept box's IoU rises to thr, or a slightly lower-scored kept box moves ahead of c
            s_iou = thr - best
            near = ks >= scores[c] - eps_s
            alt = iou_kc[near, q].max() if near.any() else -np.inf
            if alt >= thr - eps_iou and alt > best:
                need("s", "candidate %d suppressed by a kept box ranked within eps after it" % c, float((scores[c] - ks[near]).max()), eps_s)
                s_iou = max(0.0, thr - alt)
            need("i", "candidate %d (score %.7f) neither kept nor suppressed" % (c, scores[c]), s_iou, eps_iou)
    if max_keep is not None and kept.size < max_keep:
        pass        # fewer survivors than max_keep: completeness above already covered every candidate
    return rep


def summarize(rep):
    return "kept %d of %d candidates, fragile decisions %d, slack used: score %.2e, IoU %.2e%s" % (
        rep["n_kept"], rep["n_candidates"], rep["fragile"], rep["slack_score"], rep["slack_iou"],
        "" if rep["ok"] else "  VIOLATIONS: %s" % rep["violations"][:4])
