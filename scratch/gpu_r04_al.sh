#!/bin/bash
# round 4: row-per-thread Winograd transforms for launches that do not fill the chip
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
TAG=${TAG:-r04_al}
(timeout 900 python -m pytest tests/test_chain_fusion_gpu.py tests/test_dense_gpu.py tests/test_network_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v "^$" | tail -8) > gpurun_out/${TAG}_tests.txt 2>&1
(timeout 400 python -m pytest tests/test_fullsize_gpu.py -x -q -k "train or batch_invariance" 2>&1 | grep -v "^$" | tail -4) >> gpurun_out/${TAG}_tests.txt 2>&1
run() { timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 '"$*"'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))'; }
for i in 1 2; do run; run --no-fuse-chain --no-solver-in-sweep; done > gpurun_out/${TAG}_c5_ab.txt 2>&1
timeout 200 python bench.py --batch 1 --streams 1 --steps 40 --warmup 40 --no-other-configs --no-cpu-baseline --no-f32-variant 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("latency batch 1", d["value"], d["ms_per_step"])' >> gpurun_out/${TAG}_c5_ab.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_streams.py $DB gpurun_out/${TAG}_train_streams.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape.txt 10 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
cat gpurun_out/${TAG}_tests.txt gpurun_out/${TAG}_c5_ab.txt; head -12 gpurun_out/${TAG}_train_streams.txt | cut -c1-150; grep "wino4" gpurun_out/${TAG}_train_streams.txt | cut -c1-150
