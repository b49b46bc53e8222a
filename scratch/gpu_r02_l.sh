# round-2 record run (final state: streaming short-K GEMM, VGG16 / MobileNet training): GPU tests, headline bench (+ layer table), kernel trace, every BASELINE config, latency mode, PMC traffic
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02_l}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12) > gpurun_out/${TAG}_pytest.log
(timeout 300 python -m pytest tests/test_fullsize_gpu.py tests/test_boundary_gpu.py -m gpu -q -s 2>&1 | grep -E "^c[23] |max \||all-mode|passed|failed" ) > gpurun_out/${TAG}_parity_printout.txt
timeout 400 python bench.py --steps 30 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 200 python bench.py --steps 30 --warmup 5 --batch 1 --streams 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_latency.json
for cf in c1 c3 c4 c5; do timeout 300 python bench.py --config $cf --steps 12 --warmup 3 2>&1 | tail -1 > gpurun_out/${TAG}_bench_$cf.json; done
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_kernels_by_shape.txt 140 > /dev/null; find gpurun_out/prof -name '*.db' -delete
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1 )
DB=$(find gpurun_out/prof3 -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats_3chains.txt > /dev/null; find gpurun_out/prof3 -name '*.db' -delete
timeout 300 python scratch/stream_sweep.py 15,20,21,100,104,106,108 b3c1x4,b3c3x4,b4c1x4,b4c3x4,b2c1x4,b2c3x4,b1c1x4,b1c3x4,w3x4,w7x4,wrpn 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_stream_sweep.txt
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc2 gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc2 gpurun_out/prof gpurun_out/prof3
tail -4 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_parity_printout.txt | cut -c1-200; for f in gpurun_out/${TAG}_bench*.json; do echo $f; cut -c200-330 $f; done
