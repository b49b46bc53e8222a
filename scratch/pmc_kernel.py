"""Sum rocprofv3 --pmc counters per kernel-name substring: python scratch/pmc_kernel.py <dir> <substr> -> {counter: mean per launch}"""
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
