"""The decision-margin checker (oracle/margins.py) itself: it must accept exactly the greedy-NMS outcome with zero slack,
accept outcomes of slightly perturbed inputs with slack of the size of the perturbation, and reject wrong keep lists."""
import numpy as np

import frcnn_oracle as ora
import margins as mg
import synth


def _dets(k=1500, seed=7):
    return synth.random_dets(k, seed=seed, cluster=12)


def test_exact_outcome_needs_no_slack():
    d = _dets()
    keep = np.array(ora.cpu_nms(d, 0.7), dtype=np.int64)
    rep = mg.check_greedy_nms(d[:, :4], d[:, 4], keep, 0.7, 0.0, 0.0)
    assert rep["ok"] and rep["fragile"] == 0 and rep["n_kept"] == keep.size, mg.summarize(rep)
    # truncated list (proposal_layer.py:44-45) and a pre-NMS top-N (:36-38)
    order = ora.order_desc(d[:, 4])[:600]
    keep2 = order[np.array(ora.cpu_nms(d[order], 0.7), dtype=np.int64)][:50]
    rep = mg.check_greedy_nms(d[:, :4], d[:, 4], keep2, 0.7, 0.0, 0.0, topn=600, max_keep=50)
    assert rep["ok"] and rep["fragile"] == 0, mg.summarize(rep)


def test_perturbed_inputs_pass_with_small_slack():
    d = _dets(seed=11)
    rng = np.random.RandomState(0)
    p = d.copy()
    p[:, :4] += (rng.randn(d.shape[0], 4) * 2e-4).astype(np.float32)      # what f32 summation-order noise does to boxes
    p[:, 4] += (rng.randn(d.shape[0]) * 2e-7).astype(np.float32)
    keep_p = np.array(ora.cpu_nms(p, 0.7), dtype=np.int64)
    rep = mg.check_greedy_nms(d[:, :4], d[:, 4], keep_p, 0.7, 1e-5, 1e-4)
    assert rep["ok"], mg.summarize(rep)
    assert rep["slack_iou"] < 1e-4 and rep["slack_score"] < 1e-5


def test_wrong_lists_are_rejected():
    d = _dets(seed=13)
    keep = np.array(ora.cpu_nms(d, 0.7), dtype=np.int64)
    eps = dict(eps_s=1e-6, eps_iou=1e-6)
    assert not mg.check_greedy_nms(d[:, :4], d[:, 4], np.delete(keep, 3), 0.7, **eps)["ok"]          # a survivor is missing
    sup = np.setdiff1d(np.arange(d.shape[0]), keep)
    bad = np.insert(keep, 5, sup[0])
    assert not mg.check_greedy_nms(d[:, :4], d[:, 4], bad, 0.7, **eps)["ok"]                         # a suppressed box is kept
    sw = keep.copy()
    sw[[2, 40]] = sw[[40, 2]]
    assert not mg.check_greedy_nms(d[:, :4], d[:, 4], sw, 0.7, **eps)["ok"]                          # not score-ordered
    lowest = ora.order_desc(d[:, 4])[-1]
    r = mg.check_greedy_nms(d[:, :4], d[:, 4], np.append(keep[:10], lowest), 0.7, topn=200, max_keep=11, **eps)
    assert not r["ok"]                                                                               # outside the top-N cut


def test_match_to_candidates_handles_duplicates():
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [5, 5, 20, 20]], dtype=np.float32)
    scores = np.array([0.5, 0.9, 0.7], dtype=np.float32)
    m = mg.match_to_candidates(boxes[[0, 2]], np.array([0.9, 0.7]), boxes, scores, 1e-3, 1e-6)
    assert m.tolist() == [1, 2]
    m = mg.match_to_candidates(np.array([[100, 100, 120, 120]]), None, boxes, scores, 1e-3, 1e-6)
    assert m.tolist() == [-1]
