cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_s
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 500 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 300 python bench.py --config c3 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_c3.json
for f in gpurun_out/${TAG}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read()); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], (d.get('x3_variant') or {}).get('value'), (d.get('f32_mfma_variant') or {}).get('value'), r.get('frac'), r.get('mfma_busy'), r.get('traffic'), d['config'].get('detections'), d['config'].get('rois'), (d.get('cpu_baseline') or {}).get('value'))"; done
(timeout 600 python -m pytest tests/test_network_gpu.py tests/test_detect_gpu.py tests/test_boundary_gpu.py -m gpu -q 2>&1 | tail -3) > gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_pytest.log
