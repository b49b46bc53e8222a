#!/bin/bash
cd /root/repo
(timeout 300 python -m pytest tests/test_h2_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r04_aa_tests.txt 2>&1
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("telemetry"), (d.get("roofline") or {}).get("mfma_frac"))'
for i in 1 2; do for k in 1024 512; do
  timeout 300 python scratch/bench_ablation.py "-DFRCNN_H2_PP_MIN_K=$k" -- --steps 12 --warmup 4 --no-cpu-baseline --no-f32-variant --no-other-configs 2>gpurun_out/r04_aa_err.txt | python -c "$P" "pp-min-k $k" || tail -5 gpurun_out/r04_aa_err.txt
done; done > gpurun_out/r04_aa_ab_pp_min_k.txt 2>&1
cat gpurun_out/r04_aa_tests.txt gpurun_out/r04_aa_ab_pp_min_k.txt
