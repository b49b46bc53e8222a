# round 3, call b: h2 GEMM unit tests + isolated sweep vs x3 / f32, plus the re-run of the c5 train parity cases
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_b
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_h2_gpu.py -m gpu -q -x -s 2>&1 | grep -E "^h2 cfg|passed|failed|Error|error|assert" | tail -60) > gpurun_out/${TAG}_h2_tests.log
tail -30 gpurun_out/${TAG}_h2_tests.log
timeout 900 python scratch/h2_sweep.py 0,1,2,3,4,5,6 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_h2_sweep.txt
cat gpurun_out/${TAG}_h2_sweep.txt
rm -f gpurun_out/fullsize_parity.txt
(timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_detect_gpu.py -m gpu -q -k "train or detect" 2>&1 | tail -5)
cat gpurun_out/fullsize_parity.txt | cut -c1-1500
ls /sys/class/drm/ ; ls /sys/class/drm/card*/device/ | head -80; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head; ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -40
