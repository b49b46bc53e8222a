# training step (config c5): bench line + per-shape kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 300 python bench.py --config c5 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/k_bench_c5.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/k_train_by_shape.txt 10 > /dev/null; python scratch/rocpd_summary.py $DB gpurun_out/k_train_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof
cut -c1-400 gpurun_out/k_bench_c5.json; head -24 gpurun_out/k_train_kernel_stats.txt | cut -c1-60,100-175
