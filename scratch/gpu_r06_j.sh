# round 6, call J: the deferred epilogue with the scheduler free to interleave the drain pieces with the slab's MFMAs (-DFRCNN_DE_FREE_SCHED)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${TAG:-r06_j}
mkdir -p gpurun_out
echo "== fenced pieces (shipped)" > gpurun_out/${T}_h2_de_sched.txt
timeout 600 python scratch/h2_conv3.py 9,31,40 b4c3x8p,b3c3x8p,b3c3x8,w7x8,b4c3x8 >> gpurun_out/${T}_h2_de_sched.txt 2>&1
echo "== free scheduling" >> gpurun_out/${T}_h2_de_sched.txt
timeout 600 python scratch/h2_conv3.py 9,31,40 b4c3x8p,b3c3x8p,b3c3x8,w7x8,b4c3x8 --lib scratch/libfrcnn_hip_freesched.so >> gpurun_out/${T}_h2_de_sched.txt 2>&1
cat gpurun_out/${T}_h2_de_sched.txt
