cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 24 --warmup 6 --layer-report gpurun_out/layers_r1_c.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_c.json | cut -c1-1800
mkdir -p gpurun_out/prof4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof4/run.log 2>&1 )
tail -1 gpurun_out/prof4/run.log | cut -c1-200
