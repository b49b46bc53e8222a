cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 1 --no-f32-variant > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/n_by_shape.txt 10 > /dev/null
rm -rf gpurun_out/prof
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-variant --layer-report gpurun_out/n_layers.txt 2>&1 | tail -1 | cut -c1-300
head -45 gpurun_out/n_by_shape.txt | cut -c1-150
