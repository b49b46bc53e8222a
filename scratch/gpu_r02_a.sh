cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/pytest.log
POLICIES=direct,f4,f4_b1_f2,f4_b12_f2,f4_b12_direct,f4_head_f2 timeout 400 python scratch/fullsize_policies.py c2 c3 > gpurun_out/policies.log 2>&1
for v in "" "--winograd-f2 block1,block2" "--winograd-direct block1,block2" "--winograd-f2 block1,block2,block3,rpn_conv"; do
  echo "== $v" >> gpurun_out/bench_variants.log
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 $v 2>&1 | tail -1 | cut -c1-330 >> gpurun_out/bench_variants.log
done
timeout 300 python bench.py --steps 20 --warmup 5 --layer-report gpurun_out/layers.txt 2>&1 | tail -1 > gpurun_out/bench.json
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/kernel_stats.txt > /dev/null; find gpurun_out/prof -name '*.db' -delete
for cf in c3 c4 c1; do timeout 200 python bench.py --config $cf --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$cf.json; done
tail -25 gpurun_out/pytest.log; cat gpurun_out/bench_variants.log
