#!/bin/bash
cd /root/repo
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"])'
for cf in "c3 2 3" "c3 4 3" "c3 4 2" "c1 2 3" "c1 4 3" "c1 8 3" "c4 4 3" "c4 8 3" "c4 16 3"; do set -- $cf
  timeout 200 python bench.py --config $1 --batch $2 --streams $3 --steps 8 --warmup 3 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 2>/dev/null | python -c "$P" "$1 batch $2 chains $3"
done > gpurun_out/r04_ab_other_configs_batch.txt 2>&1
cat gpurun_out/r04_ab_other_configs_batch.txt
