#!/bin/bash
cd /root/repo
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("telemetry"), (d.get("roofline") or {}).get("frac"))'
for i in 1 2; do for k in 1024 512 128; do
  timeout 300 python scratch/bench_ablation.py "-DFRCNN_H2_PP_MIN_K=$k" -- --steps 12 --warmup 4 --no-cpu-baseline --no-f32-variant --no-other-configs 2>gpurun_out/r04_u_err.txt | python -c "$P" "pp-min-k $k" || tail -5 gpurun_out/r04_u_err.txt
done; done > gpurun_out/r04_u_ab_pp_min_k.txt 2>&1
cat gpurun_out/r04_u_ab_pp_min_k.txt
