# round 3: conflict-free swizzle for the 64-byte plane rows (h2 X / W planes, x3 W planes): tests, sweep, PMC, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_j
mkdir -p gpurun_out/pmc
(timeout 900 python -m pytest tests/test_h2_gpu.py tests/test_dense_gpu.py -m gpu -q -x 2>&1 | tail -4) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
timeout 900 python scratch/h2_sweep.py 0,12 b4c1x4,b4c3x4,w7x4,b3c1x4,b3c3x4,w3x4,b4c1x1 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f2)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -o p -- python $GRAFT_REPO_ROOT/scratch/h2_sweep.py 0 b4c1x4 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_kernel.py gpurun_out/pmc "k_gemm_h2<128, 128, 64, 64, 2, 2, 0>" > gpurun_out/${TAG}_pmc_gemm_h2.txt 2>&1
cat gpurun_out/${TAG}_pmc_gemm_h2.txt
rm -rf gpurun_out/pmc
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_j_bench.json").read())
print("bench:", d["value"], d["ms_per_step"], "x3:", (d.get("x3_variant") or {}).get("value"), "f32:", (d.get("f32_mfma_variant") or {}).get("value"))
r = d["roofline"]; print({k: r[k] for k in r if k not in ("kernel", "pipe_peaks_f32_equivalent")})
PY
