# round 3: epilogue with the residual in the accumulators' start value; emission A/B; small tiles; batch x chains sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_g
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_h2_gpu.py -m gpu -q -x 2>&1 | tail -5) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
timeout 900 python scratch/h2_sweep.py 0,12,13 b4c1x4,b4c3x4,w7x4,b3c1x4,b3c3x4,w3x4,b2c3x4,b2c1x4,b4c1x1,b4c3x1,w7x1,b3c1x1,b3c3x1 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
for bs in "4 3" "8 2" "8 3" "6 3" "12 1" "16 1" "2 3" "1 1"; do set -- $bs
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-f32-variant --profile-steps 0 --batch $1 --streams $2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 chains $2:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done > gpurun_out/${TAG}_batch_sweep.txt 2>&1
cat gpurun_out/${TAG}_batch_sweep.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_g_bench.json").read())
print("bench:", d["value"], d["ms_per_step"], "x3:", (d.get("x3_variant") or {}).get("value"), "f32:", (d.get("f32_mfma_variant") or {}).get("value"))
r = d["roofline"]; print({k: r[k] for k in r if k not in ("kernel", "pipe_peaks_f32_equivalent")})
PY
