cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dense_gpu.py tests/test_network_gpu.py -m gpu -q --timeout 300 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --layer-report gpurun_out/layers_r1_v2.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_v2.json
