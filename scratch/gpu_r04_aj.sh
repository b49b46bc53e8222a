#!/bin/bash
# round 4: the chain-fusion tests again + kernel trace of the training step in its final form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
TAG=${TAG:-r04_aj}
(timeout 900 python -m pytest tests/test_chain_fusion_gpu.py -q 2>&1 | grep -v "^$" | tail -8) > gpurun_out/${TAG}_tests.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_streams.py $DB gpurun_out/${TAG}_train_streams.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape.txt 10 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
cat gpurun_out/${TAG}_tests.txt; head -60 gpurun_out/${TAG}_train_streams.txt | cut -c1-150
