"""CPU: detection writers + VOC evaluator (tf-faster-rcnn_amd/lib/datasets) against the fixture that
oracle/gen_golden_eval.py produced by running the REFERENCE's own lib/datasets code on a synthetic devkit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden_eval as gge  # noqa: E402  (test infrastructure: devkit builder + fixture layout)


@pytest.fixture(scope="module")
def fixture():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "voc_eval.npz")))


def test_writers_and_voc_eval_match_reference_fixture(fixture, tmp_path):
    ve, write_voc, write_coco, _ = gge.repo_impl()
    gt, dets = fixture["gt"], fixture["dets"]
    g2, d2 = gge.synth_arrays(3, 40)
    assert np.array_equal(gt, g2) and np.array_equal(dets, d2)            # the fixture's inputs are the seeded ones
    got = gge.run(ve, write_voc, write_coco, str(tmp_path), gt, dets, 40, None)
    want = {k: v for k, v in fixture.items() if k not in ("gt", "dets")}
    assert sorted(got) == sorted(want)
    for k in want:                                                        # results files (sha), rec / prec / ap, COCO json
        assert np.asarray(got[k]).shape == want[k].shape and np.array_equal(got[k], want[k]), k
    assert 0.2 < float(want["ap_07_aeroplane"]) < 0.6                     # the synthetic case is not degenerate


def test_pascal_voc_evaluate_detections_end_to_end(fixture, tmp_path):
    from datasets.pascal_voc import pascal_voc
    gt, dets = fixture["gt"], fixture["dets"]
    devkit = tmp_path / "VOCdevkit"
    index, all_boxes, _, _ = gge.build_devkit(str(devkit / "VOC2007"), gt, dets, 40)
    imdb = pascal_voc("test", "2007", str(devkit), classes=gge.CLASSES)
    assert imdb.image_index == index and imdb.num_classes == 4
    imdb.competition_mode(True)
    aps = imdb.evaluate_detections(all_boxes, str(tmp_path / "out"), verbose=False)
    assert np.array_equal(np.array(aps), np.array([fixture["ap_07_" + c] for c in gge.CLASSES[1:]]))
    assert os.path.isfile(str(tmp_path / "out" / "bird_pr.pkl"))
    assert os.path.isfile(str(devkit / "results" / "VOC2007" / "Main" / "comp4_det_test_bird.txt"))      # competition mode keeps them
    imdb.competition_mode(False)
    imdb.evaluate_detections(all_boxes, str(tmp_path / "out"), verbose=False)
    assert len(os.listdir(str(devkit / "results" / "VOC2007" / "Main"))) == 3                            # salted files were cleaned up


def test_voc_eval_edge_cases(tmp_path):
    from datasets.voc_eval import voc_ap, voc_eval
    assert voc_ap(np.array([]), np.array([]), True) == 0 and voc_ap(np.array([]), np.array([]), False) == 0
    assert voc_ap(np.array([1.0]), np.array([1.0]), True) == pytest.approx(1.0) and voc_ap(np.array([1.0]), np.array([1.0])) == 1.0
    gt = np.array([[0, 1, 10, 10, 50, 50, 0], [0, 1, 100, 100, 150, 150, 1]], dtype=np.int64)
    dets = np.array([[1, 0, 10, 10, 50, 50, 0.9], [1, 0, 11, 11, 50, 50, 0.8], [1, 0, 100, 100, 150, 150, 0.7], [1, 1, 0, 0, 5, 5, 0.6]],
                    dtype=np.float32)
    index, all_boxes, annopath, imagesetfile = gge.build_devkit(str(tmp_path), gt, dets, 2)
    from datasets import results
    template = str(tmp_path / "det_{:s}.txt")
    results.write_voc_results_file(all_boxes, gge.CLASSES, index, template)
    assert open(template.format("aeroplane")).readline() == "000001 0.900 11.0 11.0 51.0 51.0\n"       # 1-based pixels
    rec, prec, ap = voc_eval(template, annopath, imagesetfile, "aeroplane", str(tmp_path / "cache"), use_07_metric=True)
    # hit, duplicate (fp), difficult hit (ignored), clutter on an image without objects (fp); one non-difficult positive
    assert rec.tolist() == [1.0, 1.0, 1.0, 1.0] and prec.tolist() == [1.0, 0.5, 0.5, 1.0 / 3.0] and ap == pytest.approx(1.0)
    rec, prec, ap = voc_eval(template, annopath, imagesetfile, "bird", str(tmp_path / "cache"))       # class without gt or dets
    assert rec.size == 0 and ap == 0
    rec, _, _ = voc_eval(template, annopath, imagesetfile, "aeroplane", str(tmp_path / "cache"), use_diff=True)
    assert rec.tolist() == [0.5, 0.5, 1.0, 1.0]                                                       # difficult box now counts


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib/datasets"), reason="reference tree only exists in the build container")
def test_fixture_matches_live_reference():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_eval.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "bit-exact" in r.stdout, r.stdout + r.stderr
