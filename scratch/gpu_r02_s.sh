# final check of HEAD: full GPU suite, smoke, default bench line (x3 + f32 variant + cpu_baseline), other configs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r02_w
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/${TAG}_smoke.txt
timeout 500 python bench.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 200 python bench.py --steps 30 --warmup 5 --batch 1 --streams 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_latency.json
for cf in c3; do timeout 300 python bench.py --config $cf --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_$cf.json; done
tail -2 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_smoke.txt; for f in gpurun_out/${TAG}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], (d.get('f32_mfma_variant') or {}).get('value'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
