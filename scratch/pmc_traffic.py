"""Aggregate rocprofv3 PMC passes of `bench.py` into per-launch HBM traffic of the dominant kernel family (the f32-MFMA GEMM
kernels k_conv_igemm + k_gemm_stream, i.e. every "conv:" launch of bench.py),
corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE/WRITE_SIZE are in KiB-units,
FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950 -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv, sys, json, os
root = sys.argv[1]
images = int(sys.argv[3]) if len(sys.argv) > 3 else 4            # images per launch chain of the profiled bench.py run
def avg(counter, name_filter):
    rows = list(csv.DictReader(open(os.path.join(root, counter, "p_counter_collection.csv"))))
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and any(n in r["Kernel_Name"] for n in name_filter)]
    return sum(v) / len(v), len(v)
GEMMS = ("k_conv_igemm", "k_gemm_stream", "k_gemm_x3", "k_gemm_h2")
f, n = avg("FETCH_SIZE", GEMMS)
w, _ = avg("WRITE_SIZE", GEMMS)
res = {"kernel": "k_conv_igemm + k_gemm_stream + k_gemm_x3 + k_gemm_h2", "launches": n, "FETCH_SIZE_avg": f, "WRITE_SIZE_avg": w,
       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "images_per_step": images,
       "note": "bench.py --batch %d --streams 1; (2*FETCH_SIZE + WRITE_SIZE)*1024" % images}
rows = list(csv.DictReader(open(os.path.join(root, "SQ_VALU_MFMA_BUSY_CYCLES", "p_counter_collection.csv"))))
agg = {}
for r in rows:
    if any(n in r["Kernel_Name"] for n in GEMMS):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
busy, gui = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(agg["GRBM_GUI_ACTIVE"])
res["mfma_util_conv_launches"] = busy / 1024.0 / (gui / 8.0)      # busy cycles per SIMD / kernel cycles per XCD clock
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[2], "w"), indent=1)
