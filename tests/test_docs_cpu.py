"""CPU: the evidence pointers of the documents resolve -- every `r05_*` / `r06_*` profile named in DESIGN.md / HISTORY.md / README.md /
INTEGRATION.md / bench.py / the kernels' comments exists under profiles/ (a prefix counts: `r05_g` names the files of GPU call G), the committed
counter / traffic files bench.py copies into its line have the fields it reads, and DESIGN.md stays the SHORT current design (the notebook
of rounds 1-5 is HISTORY.md)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round5_profile_references_resolve():
    names = os.listdir(os.path.join(ROOT, "profiles"))
    files = ["DESIGN.md", "HISTORY.md", "README.md", "INTEGRATION.md", "bench.py", os.path.join("scratch", "README.md"),
             os.path.join("tf-faster-rcnn_amd", "lib", "model", "config.py")]
    csrc = os.path.join("tf-faster-rcnn_amd", "csrc")
    files += [os.path.join(csrc, f) for f in sorted(os.listdir(os.path.join(ROOT, csrc))) if f.endswith((".hip", ".h"))]
    missing = []
    for f in files:
        text = open(os.path.join(ROOT, f)).read()
        for m in re.finditer(r"(r0[56]_[a-z]{1,2}_[A-Za-z0-9_]+(?:\.(?:txt|json|log))?)", text):
            ref = m.group(1)
            if ref.endswith("_"):                                  # `r05_l_*`: a call's files
                ref = ref.rstrip("_")
            if not any(n.startswith(ref) for n in names):
                missing.append((f, ref))
    assert not missing, missing


def test_committed_counter_files_have_the_fields_bench_reads():
    c = json.load(open(os.path.join(ROOT, "profiles", "r05_counters_conv3.json")))
    for k in ("source", "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES", "mfma_busy", "TCC_EA0_WRREQ_STALL/TCC_EA0_WRREQ", "SQ_LDS_BANK_CONFLICT"):
        assert k in c, k
    assert 0.0 < c["mfma_busy"] < 1.0 and 0.0 < c["SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES"] < 1.0
    t = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")))
    assert int(t["images_per_step"]) == 8 and t["hbm_bytes_per_launch"] > 1e8 and 0.0 < t["mfma_util_conv_launches"] < 1.0


def test_design_is_the_short_current_design_and_history_keeps_the_notebook():
    d = open(os.path.join(ROOT, "DESIGN.md")).read()
    h = open(os.path.join(ROOT, "HISTORY.md")).read()
    assert len(d.splitlines()) <= 300, len(d.splitlines())
    for must in ("## 1. The path and its boundary", "## 3. Data layout in HBM", "## 4. Kernels, bounds", "energy ledger", "shape_scope", "deferred epilogue"):
        assert must in d, must
    assert "Round 5 at a glance" in h and "Round 2 at a glance" in h and len(h) > 100000
    c = json.load(open(os.path.join(ROOT, "profiles", "r06_counters_conv3.json")))
    for k in ("source", "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES", "mfma_busy", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "l2_hit_rate"):
        assert k in c, k
    assert os.path.exists(os.path.join(ROOT, "profiles", "r06_energy_ledger.txt"))
