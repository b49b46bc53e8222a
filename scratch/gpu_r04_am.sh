#!/bin/bash
# round 4: 64-row tiles for batched GEMMs whose 128-row tiles would be > 25 % padding (RPN 3x3 data gradient)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-r04_am}
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_dense_gpu.py -x -q 2>&1 | grep -v "^$" | tail -4) > gpurun_out/${TAG}_tests.txt 2>&1
(timeout 400 python -m pytest tests/test_fullsize_gpu.py -x -q -k "train" 2>&1 | grep -v "^$" | tail -4) >> gpurun_out/${TAG}_tests.txt 2>&1
run() { timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]; print("c5 '"$*"'", d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step"))'; }
for i in 1 2; do run; run --no-fuse-chain --no-solver-in-sweep; done > gpurun_out/${TAG}_c5_ab.txt 2>&1
run --dp-constrained >> gpurun_out/${TAG}_c5_ab.txt 2>&1
cat gpurun_out/${TAG}_tests.txt gpurun_out/${TAG}_c5_ab.txt
