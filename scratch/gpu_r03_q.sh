cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_q
mkdir -p gpurun_out
timeout 900 python scratch/h2_sweep.py 9,19,20,21 b4c1x4,b4c3x4,w7x4,b3c1x4,b3c3x4 2>&1 | grep -v amdgpu.ids | grep -v "+p\|=p" > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
for rep in 1 2; do for args in "--h2-cfg 9" "--h2-cfg 19" "--h2-cfg 20" "--h2-cfg 21"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d.get('telemetry'))"
done; done > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
