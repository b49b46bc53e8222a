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


def test_telemetry_reports_this_process_s_card_not_the_busiest_one(tmp_path):
    """The amdgpu sysfs nodes show every card of a shared machine.  Telemetry finds the card of this process by its PCI address and reports
    the busiest OTHER card beside it (round 5: the card drawing the most power used to be reported -- a neighbour's)."""
    b = _bench()
    real = tmp_path / "pci"
    for i, (addr, mhz, watt) in enumerate((("0000:05:00.0", 2400, 1250.0), ("0000:a4:00.0", 2100, 820.0), ("0000:c4:00.0", 1900, 300.0))):
        dev = real / addr
        (dev / "hwmon" / ("hwmon%d" % i)).mkdir(parents=True)
        (dev / "pp_dpm_sclk").write_text("0: 132Mhz\n1: %dMhz *\n" % mhz)
        (dev / "hwmon" / ("hwmon%d" % i) / "power1_average").write_text(str(int(watt * 1e6)))
        (dev / "hwmon" / ("hwmon%d" % i) / "freq1_input").write_text(str(int(mhz * 1e6)))
        card = tmp_path / "drm" / ("card%d" % i)
        card.mkdir(parents=True)
        os.symlink(str(dev), str(card / "device"))
    t = b.Telemetry("0000:a4:00", root=str(tmp_path / "drm")).run(lambda: __import__("time").sleep(0.35))
    assert t["card"] == "by PCI address" and t["cards_seen"] == 3
    assert t["sclk_mhz"] == 2100 and t["socket_w"] == 820.0 and t["other_cards_max_w"] == 1250.0
    t = b.Telemetry(None, root=str(tmp_path / "drm")).run(lambda: __import__("time").sleep(0.35))        # no address: the old heuristic
    assert t["card"] != "by PCI address" and t["socket_w"] == 1250.0 and t["other_cards_max_w"] == 820.0
    assert b.Telemetry("0000:a4:00", root=str(tmp_path / "nothing")).run(lambda: None) is None


def test_train_mode_tile_threshold_rule():
    """TEST mode reads cfg.HIP.H2_MIN_TILES, TRAIN mode cfg.HIP.H2_TRAIN_MIN_TILES; None there = "the same knob as TEST" (how tests force
    frcnn_gemm_h2 onto toy TRAIN networks).  No value is compared against a default: an explicit 150 means 150."""
    from model.config import cfg
    from nets.network import Network
    keep = (cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES)
    try:
        assert keep == (150, 320)
        assert Network.h2_min_tiles("TEST") == 150 and Network.h2_min_tiles("TRAIN") == 320
        cfg.HIP.H2_MIN_TILES = 2
        assert Network.h2_min_tiles("TEST") == 2 and Network.h2_min_tiles("TRAIN") == 320
        cfg.HIP.H2_TRAIN_MIN_TILES = None
        assert Network.h2_min_tiles("TRAIN") == 2

        class Op(object):
            pass
        op = Op()
        Network.configure_train_op(op)
        assert op.h2_train == 2
        cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = 150, 100
        Network.configure_train_op(op)
        assert op.h2_train == 100 and Network.h2_min_tiles("TRAIN") == 100 and Network.h2_min_tiles("TEST") == 150
        cfg.HIP.H2_TRAIN_MIN_TILES = 150                    # an explicit 150 is 150, not "unset"
        assert Network.h2_min_tiles("TRAIN") == 150
    finally:
        cfg.HIP.H2_MIN_TILES, cfg.HIP.H2_TRAIN_MIN_TILES = keep
