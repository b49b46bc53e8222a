#!/bin/bash
# round 4: the data-gradient chain with its elementwise passes folded in (tests/test_chain_fusion_gpu.py), the training tests, C5 step time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-r04_ae}
(timeout 600 python -m pytest tests/test_chain_fusion_gpu.py -x -q 2>&1 | grep -v "^$" | tail -15) > gpurun_out/${TAG}_tests.txt 2>&1
timeout 300 python scratch/train_host_profile.py gpurun_out/${TAG}_host_profile.txt > /dev/null 2>gpurun_out/${TAG}_host_profile.err
for i in 1; do timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>gpurun_out/${TAG}_c5.err | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))'; done > gpurun_out/${TAG}_c5.txt 2>&1
timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --dp-constrained 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 dp", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))' >> gpurun_out/${TAG}_c5.txt 2>&1
cat gpurun_out/${TAG}_tests.txt gpurun_out/${TAG}_c5.txt; head -60 gpurun_out/${TAG}_host_profile.txt; tail -3 gpurun_out/${TAG}_host_profile.err; tail -5 gpurun_out/${TAG}_c5.err
