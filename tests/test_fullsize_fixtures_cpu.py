"""CPU: the committed full-size fixtures (tests/golden/full_*.npz, written by oracle/gen_fullsize.py / gen_fullsize_train.py) are
self-consistent and match the product's own variable declarations -- so the GPU box only has to run the device side."""
import numpy as np
import pytest

import fullsize as fs

CASES = [("c2", "damped"), ("c2", "calibrated"), ("c3", "calibrated"), ("c1", "damped"), ("c4", "calibrated")]


@pytest.mark.parametrize("config,weights", CASES)
def test_inference_fixture_matches_declaration_and_control(config, weights):
    c = fs.CONFIGS[config]
    net, v, image, im_info, fx = fs.build(config, weights)
    A = len(c["scales"]) * len(c["ratios"])
    Hf, Wf = int(np.ceil(c["H"] / 16.0)), int(np.ceil(c["W"] / 16.0))
    assert image.shape == (1, c["H"], c["W"], 3) and image.dtype == np.float32
    assert fx["rpn_cls_prob"].shape == (1, Hf, Wf, 2 * A) and fx["rpn_bbox_pred"].shape == (1, Hf, Wf, 4 * A)
    n = fx["rois"].shape[0]
    assert 0 < n <= c["post"] and fx["cls_prob"].shape == (n, c["classes"]) and fx["bbox_pred"].shape == (n, 4 * c["classes"])
    assert np.allclose(fx["cls_prob"].sum(axis=1), 1.0, atol=1e-5)
    # every declared variable has the declared shape after the fixture's parts were applied
    for name, sp in net.variable_specs().items():
        assert tuple(v[name].shape) == tuple(sp.shape), name
    # the class stage does representative work: no saturated ties, at most max_per_image detections
    assert fx["dets"].shape[0] <= c["max_per_image"] and np.unique(fx["dets"][:, 4]).size >= 0.9 * fx["dets"].shape[0]
    # the float32 control's tensors reproduce the control errors stored with the float64 pass
    ct = fs.load_ctrl(config, weights)
    assert ct is not None
    for k in ("rpn_cls_prob", "rpn_bbox_pred", "cls_score", "bbox_pred"):
        assert ct[k].shape == fx[k].shape
        e = fs.rel_err(ct[k], fx[k])
        assert e <= 1.2 * float(fx["ctrl_" + k]) + 1e-7, (k, e, float(fx["ctrl_" + k]))


def test_train_fixture_is_consistent():
    c = fs.TRAIN_CONFIGS["c5"]
    fx = np.load(fs.train_fixture_path("c5"))
    A = len(c["scales"]) * len(c["ratios"])
    assert fx["rois"].shape == (c["batch"], 5) and fx["pt_labels"].shape == (c["batch"], 1)
    assert fx["pt_bbox_targets"].shape == (c["batch"], 4 * c["classes"])
    assert fx["at_rpn_labels"].shape == (1, 1, A * 38, 63) and fx["at_rpn_bbox_targets"].shape == (1, 38, 63, 4 * A)
    lab = fx["at_rpn_labels"].ravel()
    assert int((lab >= 0).sum()) == 256 and int((lab == 1).sum()) <= 128            # anchor_target_layer.py:72-86
    assert int((fx["pt_labels"] > 0).sum()) <= 64                                     # proposal_target_layer.py:119-127
    for k in fs.LOSS_KEYS:
        assert np.isfinite(fx["loss_" + k]) and abs(float(fx["loss_" + k]) - float(fx["ctrl_loss_" + k])) <= 1e-4 * max(1.0, float(fx["loss_" + k]))
    net, v = fs.base_variables("c5", c["gammas"])
    fs.apply_fixture(v, net._scope, fx)
    for key, (suffix, stride) in fs.TRAIN_GRAD_SCOPES.items():
        size = int(np.prod(v[net._scope + suffix].shape))
        assert fx["grad_" + key].shape == ((size + stride - 1) // stride,) == fx["ctrl_grad_" + key].shape, key
        assert float(fx["gabs_" + key]) > 0
