cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_m
mkdir -p gpurun_out
for rep in 1 2; do
for args in "--h2-cfg 9" "--h2-cfg 18" "--h2-cfg 0"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d.get('telemetry'))"
done
done > gpurun_out/${TAG}_ab.txt 2>&1
for args in "--h2-cfg 9" "--h2-cfg 12" "--h2-cfg 17"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --batch 1 --streams 1 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch 1 x 1 chain $args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d.get('telemetry'))"
done >> gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
