# round 3: MFMA_H2 pipeline after the write-invalidation fix, per-row scale loads (any M), row-group scales in the transforms
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_f
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_h2_gpu.py tests/test_network_gpu.py -m gpu -q -x -s 2>&1 | grep -E "h2 launches|h2 path|passed|failed|Error|error|assert|fused tail" | tail -40) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
(timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "shipped and not x3 and not f32 and not train" 2>&1 | tail -5) > gpurun_out/${TAG}_tests_fullsize.log
cat gpurun_out/${TAG}_tests_fullsize.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_f_bench.json").read())
print("bench:", d["value"], d["ms_per_step"], "x3:", (d.get("x3_variant") or {}).get("value"), "f32:", (d.get("f32_mfma_variant") or {}).get("value"))
r = d["roofline"]; print({k: r[k] for k in r if k not in ("kernel", "pipe_peaks_f32_equivalent")})
print(json.dumps(d.get("stages")))
PY
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-variant --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_kernels_by_shape.txt 140 > /dev/null; find gpurun_out/prof -name '*.db' -delete
head -40 gpurun_out/${TAG}_kernels_by_shape.txt
rm -rf gpurun_out/prof
