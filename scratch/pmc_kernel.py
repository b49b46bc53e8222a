"""Sum rocprofv3 --pmc counters per kernel-name substring: python scratch/pmc_kernel.py <dir> <substr> [out.json [note]] -> {counter: mean per launch}
(out.json: the ratios bench.py copies into `roofline.counters`)"""
import csv, glob, os, sys
root, sub = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-32s launches %4d  mean %.4g" % (k, len(v), sum(v) / len(v)))
if "SQ_WAVE_CYCLES" in acc:
    wc = sum(acc["SQ_WAVE_CYCLES"])
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU",
              "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT"):
        if k in acc:
            print("%-32s / SQ_WAVE_CYCLES = %.3f" % (k, sum(acc[k]) / wc))

if len(sys.argv) > 3:
    import json
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    wc = mean.get("SQ_WAVE_CYCLES")
    out = {"source": sys.argv[4] if len(sys.argv) > 4 else root, "launches": len(acc.get("SQ_WAVE_CYCLES", [])),
           "counters_mean_per_launch": {k: round(v, 1) for k, v in sorted(mean.items())}}
    if wc:
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in mean:
                out[k + "/SQ_WAVE_CYCLES"] = round(mean[k] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        # busy cycles per SIMD (256 CUs x 4) / kernel cycles per XCD clock (GRBM_GUI_ACTIVE sums the 8 XCDs): scratch/pmc_traffic.py's formula
        out["mfma_busy"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if mean.get("TCC_EA0_WRREQ"):
        out["TCC_EA0_WRREQ_STALL/TCC_EA0_WRREQ"] = round(mean.get("TCC_EA0_WRREQ_STALL", 0.0) / mean["TCC_EA0_WRREQ"], 4)
    if "SQ_LDS_BANK_CONFLICT" in mean:
        out["SQ_LDS_BANK_CONFLICT"] = mean["SQ_LDS_BANK_CONFLICT"]
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        out["hbm_bytes_per_launch"] = round((2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0)      # the guide's gfx950 corrections
    json.dump(out, open(sys.argv[3], "w"), indent=1)
