# round 3, first pipeline run with cfg.HIP.MFMA_H2: h2 + network + full-size parity, bench with variants, 16-bit MFMA ceiling, alignment probe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_e
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_h2_gpu.py tests/test_network_gpu.py tests/test_dense_gpu.py -m gpu -q -x -s 2>&1 | grep -E "h2 launches|h2 path|passed|failed|Error|error|assert|fused tail" | tail -40) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
(timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_tests_fullsize.log
cat gpurun_out/${TAG}_tests_fullsize.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
timeout 120 python scratch/h2_unaligned.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_unaligned.txt; cat gpurun_out/${TAG}_h2_unaligned.txt
timeout 600 python bench.py --steps 30 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_e_bench.json").read())
print("bench:", d["value"], d["ms_per_step"], "x3:", (d.get("x3_variant") or {}).get("value"), "f32:", (d.get("f32_mfma_variant") or {}).get("value"))
print(json.dumps(d.get("roofline"), indent=0)[:1800])
print(json.dumps(d.get("stages"))[:1500])
PY
hipcc -O3 --offload-arch=gfx950 scratch/mfma_peak16.hip -o /tmp/mfma_peak16 && timeout 120 /tmp/mfma_peak16 > gpurun_out/${TAG}_mfma_peak16.txt; cat gpurun_out/${TAG}_mfma_peak16.txt
