# streaming short-K GEMM: tests that touch the dense path, A/B bench, per-shape kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/h_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/h_bench.json
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stream-gemm 2>&1 | tail -1 > gpurun_out/h_bench_nostream.json
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/h_bench2.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/h_by_shape.txt 330 > /dev/null
rm -rf gpurun_out/prof
tail -3 gpurun_out/h_pytest.log; for f in gpurun_out/h_bench.json gpurun_out/h_bench_nostream.json gpurun_out/h_bench2.json; do python -c "
import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('frac_launched'))"; done; head -30 gpurun_out/h_by_shape.txt | cut -c1-150
