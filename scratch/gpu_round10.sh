cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --layer-report gpurun_out/layers_r1_d.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_d.json | cut -c1-300
mkdir -p gpurun_out/prof5
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof5/run.log 2>&1 )
tail -1 gpurun_out/prof5/run.log | cut -c1-200
DB=$(find gpurun_out/prof5 -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/r01_d_kernel_stats.txt | cut -c1-60,100-190 | head -24; find gpurun_out/prof5 -name '*.db' -delete
