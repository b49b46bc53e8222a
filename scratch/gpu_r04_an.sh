#!/bin/bash
# round 4: where do the 6 ms of the data-parallel rules go?  kernel trace of bench.py --config c5 --dp-constrained, per stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
TAG=${TAG:-r04_an}
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --dp-constrained > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_streams.py $DB gpurun_out/${TAG}_train_streams_dp.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape_dp.txt 10 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
head -14 gpurun_out/${TAG}_train_streams_dp.txt | cut -c1-150; grep -i "ccl\|copyBuffer\|sgd" gpurun_out/${TAG}_train_kernels_by_shape_dp.txt | cut -c1-160
