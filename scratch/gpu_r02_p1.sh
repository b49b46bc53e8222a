cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/p1_pytest.log
for cf in c1 c3 c4; do timeout 300 python bench.py --config $cf --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/p1_bench_$cf.json; done
tail -6 gpurun_out/p1_pytest.log; for cf in c1 c3 c4; do python -c "
import json; d=json.loads(open('gpurun_out/p1_bench_$cf.json').read()); print('$cf', d['value'], d['ms_per_step'], d.get('f32_mfma_variant',{}).get('value'))"; done
