# round 5, call Q: per-stream view of the replayed training step at the round-5 state (profiles/r04_al_train_streams.txt was round 4's)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_q}
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 3 --no-other-configs --no-cpu-baseline --pick-streams 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_streams.py $DB gpurun_out/${TAG}_train_streams.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape.txt 10 > /dev/null; find gpurun_out/prof -name '*.db' -delete
tail -2 gpurun_out/prof/run.log | cut -c1-300; rm -rf gpurun_out/prof
head -60 gpurun_out/${TAG}_train_streams.txt | cut -c1-170
