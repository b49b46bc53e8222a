cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/scratch/c5_fwd.py > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 ); tail -1 gpurun_out/prof/run.log
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/r03_ak_fwd_kernels_by_shape.txt 70 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
head -5 gpurun_out/r03_ak_fwd_kernels_by_shape.txt | cut -c1-100
