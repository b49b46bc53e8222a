cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_u
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_h2_gpu.py -m gpu -q -x -k "22 or planes or residual or chained" 2>&1 | tail -4) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
timeout 900 python scratch/h2_sweep.py 9,22 b4c1x4,b4c3x4,w7x4,wrpn,b3c1x4,b3c3x4,w3x4,b4c1x1 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
timeout 600 python scratch/h2_trace.py 22 2>&1 | grep -v "amdgpu.ids\|warning" > gpurun_out/${TAG}_h2_trace_cfg22.txt; head -12 gpurun_out/${TAG}_h2_trace_cfg22.txt
for rep in 1 2; do for args in "--h2-cfg 9" "--h2-cfg 22"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d.get('telemetry'))"
done; done > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
