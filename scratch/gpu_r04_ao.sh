#!/bin/bash
# round 4: what do the data-parallel rules cost at N = 1, piece by piece (same box, back to back)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-r04_ao}
run() { timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 '"$*"'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"), c.get("all_reduce_host_ms_per_step"))'; }
(run --wgrad-streams 1 --no-solver-in-sweep; run --dp-constrained --dp-probe noop-nogroup; run --dp-constrained --dp-probe noop) > gpurun_out/${TAG}_dp_pieces2.txt 2>&1
cat gpurun_out/${TAG}_dp_pieces2.txt
