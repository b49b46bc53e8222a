#!/bin/bash
# r04_o: fused-mean tests + A/B, the dp-constrained training step (stderr kept), batch-8 default
cd /root/repo
(timeout 600 python -m pytest tests/test_h2_gpu.py tests/test_dense_gpu.py tests/test_network_gpu.py -x -q -k "mean or batched or stream" 2>&1 | tail -8) > gpurun_out/r04_o_tests.txt 2>&1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("telemetry"), (d.get("roofline") or {}).get("frac"))'
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-other-configs --dp-constrained > gpurun_out/r04_o_c5dp.out 2> gpurun_out/r04_o_c5dp.err; tail -c 600 gpurun_out/r04_o_c5dp.out | python -c "$P" "c5 dp" || tail -20 gpurun_out/r04_o_c5dp.err
for i in 1 2; do for f in "" "--fused-mean"; do
  timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-f32-variant --no-other-configs $f 2>/dev/null | python -c "$P" "[$f]"
done; done > gpurun_out/r04_o_ab_mean.txt 2>&1
cat gpurun_out/r04_o_tests.txt gpurun_out/r04_o_ab_mean.txt
