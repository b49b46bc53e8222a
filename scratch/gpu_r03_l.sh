cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_l
mkdir -p gpurun_out
for rep in 1 2; do
for args in "--h2-cfg -1" "--h2-cfg 0" "--h2-cfg 9" "--h2-cfg 0 --h2-trunk-planes 1" "--h2-cfg 9 --h2-trunk-planes 1" "--mfma x3"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 $args 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done
done > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
