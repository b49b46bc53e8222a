#!/bin/bash
cd /root/repo
(timeout 300 python -m pytest tests/test_network_gpu.py tests/test_fullsize_gpu.py -x -q -k "direct_conv or fused or batch_invariance or criterion" 2>&1 | grep -v "^$" | tail -8) > gpurun_out/r04_t_tests.txt 2>&1
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-other-configs --dp-constrained 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 dp", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"), c.get("all_reduce_host_ms_per_step"), c.get("all_reduce_calls_per_step"))' > gpurun_out/r04_t_c5dp.txt 2>&1
cat gpurun_out/r04_t_tests.txt gpurun_out/r04_t_c5dp.txt
