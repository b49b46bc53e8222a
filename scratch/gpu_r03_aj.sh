cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_aj
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_train_dp_gpu.py -m gpu -q 2>&1 | tail -4) > gpurun_out/${TAG}_dp_tests.log; cat gpurun_out/${TAG}_dp_tests.log
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape.txt 70 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
head -12 gpurun_out/${TAG}_train_kernels_by_shape.txt | cut -c1-140
