cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -2
timeout 600 python bench.py --steps 60 --warmup 6 --layer-report gpurun_out/layers_r1_final.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_final.json | cut -c1-1500
mkdir -p gpurun_out/prof2
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof2/run.log 2>&1 )
ls gpurun_out/prof2
