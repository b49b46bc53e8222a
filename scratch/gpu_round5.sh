cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3
for s in 1 2 3; do timeout 300 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --streams $s --layer-report gpurun_out/layers_r1_v3_s$s.txt 2>&1 | tail -1 | tee gpurun_out/bench_r1_v3_s$s.json | cut -c1-330; done
