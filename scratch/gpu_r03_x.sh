cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_x
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "mobilenet" 2>&1 | tail -30) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
