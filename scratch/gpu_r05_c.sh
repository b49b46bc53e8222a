# round 5, call C: phase-by-phase s_memtime trace of the tile boundary, cfg 9 vs the light-boundary forms
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scratch/h2_trace_boundary.py ${CFGS:-9,30,31,32} ${SHAPES:-b3c3x8,b4c3x8} > gpurun_out/${TAG:-r05_c}_h2_trace_boundary.txt 2>&1
cat gpurun_out/${TAG:-r05_c}_h2_trace_boundary.txt
