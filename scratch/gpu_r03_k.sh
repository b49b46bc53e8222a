#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ag; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/train_tests.txt 2>&1; tail -3 $O/train_tests.txt
TN=1 SIDE=2 python scratch/c5_phases.py 2>&1 | tail -4
for a in "" ""; do timeout 300 python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline $a 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])"; done
