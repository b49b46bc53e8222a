#!/bin/bash
# wgrad side stream: tests + c5 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_aa
O=gpurun_out/r03_aa
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/train_tests.txt 2>&1
tail -5 $O/train_tests.txt
for i in 1 2; do
timeout 300 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline > $O/c5_side_$i.json 2> $O/c5_side_$i.err
timeout 300 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline --no-wgrad-stream > $O/c5_main_$i.json 2> $O/c5_main_$i.err
done
for f in $O/c5_*.json; do echo $f; python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
except Exception as e: print("ERR", e)
PY
done
