# validation of the parallel bin selection + per-shape kernel trace of the graph-replayed step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/g_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/g_bench.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/g_by_shape.txt 330 > /dev/null; python scratch/rocpd_summary.py $DB gpurun_out/g_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof
tail -3 gpurun_out/g_pytest.log; cut -c1-400 gpurun_out/g_bench.json; head -50 gpurun_out/g_by_shape.txt
