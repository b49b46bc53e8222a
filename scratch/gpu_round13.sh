cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_train
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o t -- python $GRAFT_REPO_ROOT/scratch/train_speed.py 152 > $GRAFT_REPO_ROOT/gpurun_out/prof_train/run.log 2>&1 )
tail -2 gpurun_out/prof_train/run.log | cut -c1-200
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/r01_e_train_kernel_stats.txt | cut -c1-70,100-190 | head -30; find gpurun_out/prof_train -name '*.db' -delete
