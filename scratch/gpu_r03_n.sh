cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_n
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_h2_gpu.py -m gpu -q -x 2>&1 | tail -6) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
timeout 900 python scratch/h2_sweep.py 9,20,22 b4c1x4,b4c3x4,w7x4,wrpn,b3c1x4,b3c3x4,w3x4 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
