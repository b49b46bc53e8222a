# round-2 record run of the FINAL state (cfg.HIP.MFMA_X3 on): tests, parity printouts, bench (x3 + f32 variant in the same run), all configs,
# kernel traces, x3 sweep, clock / power samples, PMC passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r02_p
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12) > gpurun_out/${TAG}_pytest.log
(timeout 400 python -m pytest tests/test_fullsize_gpu.py tests/test_boundary_gpu.py tests/test_dense_gpu.py -m gpu -q -s -k "fullsize or boundary or gemm_x3" 2>&1 | grep -E "^c[23] |max \||all-mode|x3 \(|passed|failed" | cut -c1-330 ) > gpurun_out/${TAG}_parity_printout.txt
timeout 500 python bench.py --steps 30 --warmup 5 --layer-report gpurun_out/${TAG}_layer_table.txt 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
timeout 200 python bench.py --steps 30 --warmup 5 --batch 1 --streams 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_latency.json
for cf in c1 c3 c4 c5; do timeout 300 python bench.py --config $cf --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_$cf.json; done
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-variant --profile-steps 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null; python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_kernels_by_shape.txt 140 > /dev/null; find gpurun_out/prof -name '*.db' -delete
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-f32-variant --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1 )
DB=$(find gpurun_out/prof3 -name '*.db' | head -1); python scratch/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats_3chains.txt > /dev/null; find gpurun_out/prof3 -name '*.db' -delete
timeout 300 python scratch/x3_sweep.py 0,1,10,11,12 small,b4c1x4,b4c3x4,w7x4,wrpn,b3c1x4,b3c3x4,w3x4,b3c1x1,b3c3x1,w3x1,b4c1x1,b4c3x1,w7x1 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_x3_sweep.txt
for M in x3 f32; do
  ( for i in $(seq 1 44); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/p_smi_$M.txt &
  SMI=$!
  timeout 300 python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --mfma $M 2>&1 | tail -1 | grep -o '"value": [0-9.]*, "ms_per_step": [0-9.]*' > gpurun_out/p_bench_$M.txt
  wait $SMI
  echo "== bench.py --mfma $M --steps 600: $(cat gpurun_out/p_bench_$M.txt)"
  echo "sclk MHz while busy: $(grep -o 'sclk clock level: [0-9S]: ([0-9]*Mhz)' gpurun_out/p_smi_$M.txt | grep -o '([0-9]*' | tr -d '(' | awk '$1>1000' | sort -n | awk '{a[NR]=$1} END{print "min",a[1],"median",a[int((NR+1)/2)],"max",a[NR],"n",NR}')"
  echo "socket power W while busy: $(grep -o 'Power (W): [0-9.]*' gpurun_out/p_smi_$M.txt | awk '{print $3}' | awk '$1>900' | sort -n | awk '{a[NR]=$1} END{print "min",a[1],"median",a[int((NR+1)/2)],"max",a[NR],"n",NR}')"
done > gpurun_out/${TAG}_clock_power.txt
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline --no-f32-variant > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc2 gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc2 gpurun_out/prof gpurun_out/prof3
tail -3 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_clock_power.txt; for f in gpurun_out/${TAG}_bench*.json; do python -c "
import json; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], (d.get('f32_mfma_variant') or {}).get('value'), (d.get('roofline') or {}).get('frac'))"; done; tail -5 gpurun_out/pmc_traffic.log
