# round-3 closing run after the training-step work: full GPU suite, smoke, default bench, c5 bench (+ side-stream count A/B)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r03_am}
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt
timeout 500 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 300 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_c5.json
for n in 1 2 3; do timeout 300 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline --wgrad-streams $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgrad side streams $n:', d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"; done > gpurun_out/${TAG}_c5_streams.txt; cat gpurun_out/${TAG}_c5_streams.txt
for f in gpurun_out/${TAG}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read()); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], (d.get('x3_variant') or {}).get('value'), (d.get('f32_mfma_variant') or {}).get('value'), r.get('frac'))"; done
