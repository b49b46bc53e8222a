cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_v
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_h2_gpu.py tests/test_network_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "depthwise or mobilenet or vgg or c4" 2>&1 | tail -4) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
grep "^c4" gpurun_out/fullsize_parity.txt | cut -c1-330
for rep in 1 2; do
timeout 300 python bench.py --config c4 --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; x3', (d.get('x3_variant') or {}).get('value'), 'f32', (d.get('f32_mfma_variant') or {}).get('value'))"
done > gpurun_out/${TAG}_c4.txt 2>&1; cat gpurun_out/${TAG}_c4.txt
