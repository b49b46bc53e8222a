#!/bin/bash
# round 4: do the streams of a training step share hardware queues?  GPU_MAX_HW_QUEUES (ROCm CLR, default 4) A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-r04_ap}
run() { timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 '"$*"' GPU_MAX_HW_QUEUES='$GPU_MAX_HW_QUEUES'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))'; }
(run; run --dp-constrained; export GPU_MAX_HW_QUEUES=8; run; run --dp-constrained) > gpurun_out/${TAG}_hw_queues.txt 2>&1
cat gpurun_out/${TAG}_hw_queues.txt
