cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scratch/h2_trace.py 9 2>&1 | grep -v "amdgpu.ids\|warning" > gpurun_out/r03_t_h2_trace.txt; cat gpurun_out/r03_t_h2_trace.txt
