cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_c
mkdir -p gpurun_out
python scratch/h2_debug.py 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-200
(timeout 600 python -m pytest tests/test_h2_gpu.py -m gpu -q -s 2>&1 | grep -E "^h2 cfg|passed|failed|Error|error|assert" | tail -60) > gpurun_out/${TAG}_h2_tests.log
tail -40 gpurun_out/${TAG}_h2_tests.log
timeout 900 python scratch/h2_sweep.py 0,1,2,3,4,5,6 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
