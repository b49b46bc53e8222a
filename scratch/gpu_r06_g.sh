# round 6, call G: the Winograd transforms, row-per-thread threshold sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${TAG:-r06_g}
mkdir -p gpurun_out
timeout 600 python scratch/wino_bench.py 256,512,100000000 > gpurun_out/${T}_wino_bench.txt 2>&1; cat gpurun_out/${T}_wino_bench.txt
