set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_r1_a.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-graph --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_r1_eager.json
