# round-6 record run: full GPU suite, parity printouts, the default bench line (h2 + x3 + f32 variants, other_configs, latency, cpu_baseline),
# smoke, kernel trace, PMC passes (HBM traffic / matrix-pipe busy of the GEMM launches at the shipped batch of 8)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06_z}
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
(timeout 300 python -m pytest tests/test_detect_gpu.py tests/test_network_gpu.py tests/test_boundary_gpu.py tests/test_h2_gpu.py -m gpu -q -s -k "cuda_kernel or h2_path or other_modes or bbox_reg or mean" 2>&1 | grep -E "kept|h2 launches|h2 path|gemm_h2_mean|ulp|passed|failed" | cut -c1-300) > gpurun_out/${TAG}_printouts.txt
cat gpurun_out/${TAG}_printouts.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json ) 2> gpurun_out/${TAG}_bench_wall.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_kernels_by_shape.txt 140 > /dev/null; find gpurun_out/prof -name '*.db' -delete
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc2 gpurun_out/${TAG}_pmc_traffic.json 8 > gpurun_out/pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc2 gpurun_out/prof
tail -3 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_bench_wall.txt | tail -4; python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], (d.get('x3_variant') or {}).get('value'), (d.get('f32_mfma_variant') or {}).get('value'), r.get('frac'), r.get('sclk_mhz'), r.get('socket_w'), d.get('latency_ms_batch1')); print({k: (v.get('value'), v.get('ms_per_step'), v.get('roofline_frac')) for k, v in (d.get('other_configs') or {}).items()})"; tail -8 gpurun_out/pmc_traffic.log
