cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --layer-report gpurun_out/layers.txt 2>&1 | tail -1 > gpurun_out/bench.json
timeout 200 python bench.py --steps 20 --warmup 5 --batch 1 --streams 1 --no-cpu-baseline --profile-steps 0 2>&1 | tail -1 | cut -c1-400 > gpurun_out/bench_latency.json
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/kernel_stats.txt > /dev/null; find gpurun_out/prof -name '*.db' -delete
timeout 300 python bench.py --config c5 --steps 8 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_c5.json
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc2 gpurun_out/pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
find gpurun_out/pmc2 -name '*.csv' -size +3M -delete
tail -30 gpurun_out/pytest.log; cut -c1-300 gpurun_out/bench.json; cat gpurun_out/bench_latency.json; cut -c1-300 gpurun_out/bench_c5.json; tail -12 gpurun_out/pmc_traffic.log
