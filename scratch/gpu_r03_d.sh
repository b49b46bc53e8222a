cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_d
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_h2_gpu.py -m gpu -q 2>&1 | tail -5) > gpurun_out/${TAG}_h2_tests.log
tail -5 gpurun_out/${TAG}_h2_tests.log
timeout 900 python scratch/h2_sweep.py 0,5,7,8,9,10,11 b4c1x4,b4c3x4,w7x4,b3c1x4,b3c3x4,w3x4,b2c3x4 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
