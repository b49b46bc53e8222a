cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01_f}
timeout 600 python bench.py --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
mkdir -p gpurun_out/prof_$TAG
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/run.log 2>&1 )
DB=$(find gpurun_out/prof_$TAG -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt | cut -c1-60,100-190 | head -12; find gpurun_out/prof_$TAG -name '*.db' -delete
timeout 300 python scratch/run_configs.py 2>&1 | grep -E "^C[1-5]" | tee gpurun_out/${TAG}_all_configs.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_$TAG -o t -- python $GRAFT_REPO_ROOT/scratch/train_speed.py 152 > $GRAFT_REPO_ROOT/gpurun_out/prof_train_$TAG.log 2>&1 )
tail -1 gpurun_out/prof_train_$TAG.log | cut -c1-200
DB=$(find gpurun_out/prof_train_$TAG -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_train_kernel_stats.txt > /dev/null; find gpurun_out/prof_train_$TAG -name '*.db' -delete
