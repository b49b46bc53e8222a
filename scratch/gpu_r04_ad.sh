#!/bin/bash
# round 4: which chain of the training step is the long one (per-stream view of a kernel trace) + side-stream count A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
TAG=${TAG:-r04_ad}
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_streams.py $DB gpurun_out/${TAG}_train_streams.txt > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
for w in 0 1 2 3; do timeout 200 python bench.py --config c5 --steps 15 --warmup 4 --no-other-configs --no-cpu-baseline --wgrad-streams $w 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 wgrad-streams '$w'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))'; done > gpurun_out/${TAG}_c5_streams.txt 2>&1
cat gpurun_out/${TAG}_train_streams.txt gpurun_out/${TAG}_c5_streams.txt
