#!/bin/bash
cd /root/repo
(timeout 400 python -m pytest tests/test_h2_gpu.py tests/test_network_gpu.py -x -q -k "mean or batched" 2>&1 | grep -v "^$" | tail -30) > gpurun_out/r04_q_tests.txt 2>&1
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("telemetry"), (d.get("roofline") or {}).get("frac"), {k: v["us_per_image"] for k, v in (d.get("stages") or {}).items() if "mean" in k})'
for i in 1 2; do for f in "" "--fused-mean"; do
  timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-f32-variant --no-other-configs $f 2>gpurun_out/r04_q_err.txt | python -c "$P" "[$f]" || tail -5 gpurun_out/r04_q_err.txt
done; done > gpurun_out/r04_q_ab_mean.txt 2>&1
cat gpurun_out/r04_q_tests.txt gpurun_out/r04_q_ab_mean.txt
