cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_network_gpu.py -m gpu -q --timeout 300 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --layer-report gpurun_out/layers_r1.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_b.json
mkdir -p gpurun_out/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
ls -R gpurun_out/prof | head -30
