#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ai; mkdir -p $O
rm -f gpurun_out/fullsize_parity.txt
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_wgrad_gpu.py -x -q -m gpu > $O/train_tests.txt 2>&1; tail -3 $O/train_tests.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "train" > $O/fullsize_train.txt 2>&1; tail -2 $O/fullsize_train.txt
for i in 1 2 3; do for a in "" "--no-h2-train-wino"; do python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$a\", d[\"value\"], d[\"ms_per_step\"])"; done; done
