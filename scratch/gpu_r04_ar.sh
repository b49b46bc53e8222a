#!/bin/bash
# round 4: does the existence of an RCCL group change the inference rate per GPU?  (stream pool -> hardware queues; N = 1, no collective)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 60 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-f32-variant --profile-steps 0 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("c2 '"$*"'", d["value"], d["ms_per_step"])'; }
(run; run --one-rank-group) > gpurun_out/r04_ar_group_inference.txt 2>&1
cat gpurun_out/r04_ar_group_inference.txt
