cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_r
mkdir -p gpurun_out
for rep in 1 2; do for args in "--h2-cfg 9" "--h2-cfg 3" "--h2-cfg 4" "--h2-cfg 1"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d.get('telemetry'))"
done; done > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^$" | head -30 > gpurun_out/${TAG}_smi.txt; cat gpurun_out/${TAG}_smi.txt | head -20
