"""CPU: bench.py keeps the driver's contract without a GPU -- flags the driver passes exist, the metric string and the workloads are
BASELINE.json's, the constants the roofline is priced against are the guide's."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_metric_configs_and_peaks_match_baseline():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert b.METRIC == base["metric"]
    assert sorted(b.CONFIGS) == ["c1", "c2", "c3", "c4", "c5"] and len(base["configs"]) == 5
    c2 = b.CONFIGS["c2"]
    assert (c2["H"], c2["W"], c2["post"], c2["classes"]) == (600, 1000, 300, 21) and c2["net"] == "res101"      # configs[1], the metric's
    c3 = b.CONFIGS["c3"]
    assert (c3["H"], c3["W"], c3["post"], c3["classes"]) == (800, 1333, 1000, 81) and len(c3["scales"]) == 5
    assert b.F32_MFMA_PEAK_TFLOPS == 157.3 and abs(b.X3_PEAK_TFLOPS - 2500.0 / 6.0) < 0.1 and b.HBM_PEAK_GBS == 8000.0
    assert abs(b.H2_PEAK_TFLOPS - 2500.0 / 3.0) < 0.1


def test_driver_flags_and_variants_parse():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--mfma", "--dp-constrained", "--no-other-configs", "--no-f32-variant", "--no-cpu-baseline", "--layer-report"):
        assert '"%s"' % flag in src, flag
    # the shipped configuration and its x3-only / all-f32-MFMA twins are all reported by the default invocation
    assert 'default="h2"' in src and '"x3_variant"' in src and '"f32_mfma_variant"' in src and '"cpu_baseline"' in src and '"roofline"' in src
    assert "RANK" in src and "WORLD_SIZE" in src and "dist.barrier()" in src and "torch.cuda.synchronize()" in src
