cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --no-cpu-baseline --profile-steps 0 --streams 3 > $GRAFT_REPO_ROOT/gpurun_out/prof3/run.log 2>&1 )
tail -1 gpurun_out/prof3/run.log | cut -c1-200
ls -la gpurun_out/prof3
