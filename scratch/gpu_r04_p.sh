#!/bin/bash
cd /root/repo
(timeout 300 python -m pytest tests/test_h2_gpu.py -x -q -k "mean" 2>&1 | grep -v "^$" | head -60) > gpurun_out/r04_p_tests.txt 2>&1
PYTHONFAULTHANDLER=1 timeout -s ABRT 70 python bench.py --config c5 --steps 10 --warmup 3 --no-other-configs --dp-constrained > gpurun_out/r04_p_c5dp.out 2> gpurun_out/r04_p_c5dp.err
tail -c 400 gpurun_out/r04_p_c5dp.out; grep -v "^$" gpurun_out/r04_p_c5dp.err | tail -60
cat gpurun_out/r04_p_tests.txt
