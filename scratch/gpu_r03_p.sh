# round-3 record run: full GPU suite, parity printouts, bench (h2 + x3 + f32 variants in one run), latency, other configs, kernel trace, PMC
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r03_p}
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
(timeout 300 python -m pytest tests/test_detect_gpu.py tests/test_network_gpu.py tests/test_boundary_gpu.py -m gpu -q -s -k "cuda_kernel or h2_path or other_modes or bbox_reg" 2>&1 | grep -E "kept|h2 launches|h2 path|passed|failed" | cut -c1-300) > gpurun_out/${TAG}_printouts.txt
cat gpurun_out/${TAG}_printouts.txt
timeout 500 python bench.py --steps 30 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 200 python bench.py --steps 30 --warmup 5 --batch 1 --streams 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_latency.json
for cf in c1 c3 c4 c5; do timeout 300 python bench.py --config $cf --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_$cf.json; done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-variant --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_kernels_by_shape.txt 140 > /dev/null; find gpurun_out/prof -name '*.db' -delete
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline --no-f32-variant > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc2 gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc2 gpurun_out/prof
tail -3 gpurun_out/${TAG}_pytest.log; for f in gpurun_out/${TAG}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read()); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], (d.get('x3_variant') or {}).get('value'), (d.get('f32_mfma_variant') or {}).get('value'), r.get('frac'), r.get('sclk_mhz'), r.get('socket_w'), d['config'].get('detections'))"; done; tail -8 gpurun_out/pmc_traffic.log
