cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_w
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_dp_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "train" 2>&1 | tail -4) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
grep -A4 "TRAIN" gpurun_out/fullsize_parity.txt | cut -c1-400
for h in 1 0; do
timeout 300 python - <<PY
import sys, json, subprocess
sys.argv=['x']
PY
done
timeout 300 python bench.py --config c5 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 (H2_TRAIN default):', d['value'], 'steps/s', d['ms_per_step'], 'ms/step')" > gpurun_out/${TAG}_c5.txt
cat gpurun_out/${TAG}_c5.txt
