#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ae; mkdir -p $O
timeout 600 python -m pytest tests/test_wgrad_gpu.py -x -q -m gpu -s > $O/wgrad_tests.txt 2>&1
grep -E "wgrad h2.*err|passed|failed|Error|error" $O/wgrad_tests.txt | tail -28
timeout 300 python scratch/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_bench.txt
for a in "" "--no-wgrad-h2"; do timeout 300 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline $a 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])"; done
