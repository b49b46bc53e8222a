#!/bin/bash
# r04_n: tests after the prune, the training step with / without the data-parallel rules, trunk planes A/B, batch x chains sweep
cd /root/repo
(timeout 600 python -m pytest tests/test_h2_gpu.py tests/test_train_gpu.py tests/test_train_dp_gpu.py tests/test_wgrad_gpu.py -x -q 2>&1 | tail -6) > gpurun_out/r04_n_tests.txt 2>&1
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d.get("telemetry"), (d.get("roofline") or {}).get("frac"))'
for f in "" "--dp-constrained"; do
  timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-other-configs $f 2>/dev/null | python -c "$P" "c5 [$f]"
done > gpurun_out/r04_n_c5.txt 2>&1
for i in 1 2; do for t in 0 1; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-variant --no-other-configs --h2-trunk-planes $t 2>/dev/null | python -c "$P" "trunk-planes $t"
done; done > gpurun_out/r04_n_ab_trunk.txt 2>&1
for bs in "4 3" "6 3" "8 3" "8 2" "12 2"; do set -- $bs
  timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --batch $1 --streams $2 2>/dev/null | python -c "$P" "batch $1 chains $2"
done > gpurun_out/r04_n_batch_sweep.txt 2>&1
cat gpurun_out/r04_n_tests.txt gpurun_out/r04_n_c5.txt gpurun_out/r04_n_ab_trunk.txt gpurun_out/r04_n_batch_sweep.txt
