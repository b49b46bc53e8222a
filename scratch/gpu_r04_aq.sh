#!/bin/bash
# round 4: filter-gradient side streams chosen by a concurrency probe (TrainState._concurrent_stream)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-r04_aq}
run() { timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" 2>gpurun_out/${TAG}.err | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 '"$*"'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"), "")'; }
(run; run --dp-constrained) > gpurun_out/${TAG}_head.txt 2>&1
cat gpurun_out/${TAG}_head.txt; tail -3 gpurun_out/${TAG}.err
