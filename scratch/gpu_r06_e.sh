# round 6, call E: the deferred epilogue's tests, the energy ledger, the pipeline A/B of the DE dispatch rule (-7 = round 5's choice)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${TAG:-r06_e}
timeout 1200 python -m pytest tests/test_h2_gpu.py tests/test_chain_fusion_gpu.py -q -x > gpurun_out/${T}_h2_tests.txt 2>&1; tail -6 gpurun_out/${T}_h2_tests.txt
B="--no-cpu-baseline --no-other-configs --no-f32-variant --profile-steps 0 --steps 40"
for i in 1 2; do
  timeout 300 python bench.py $B --hip H2_TILE_CFG=-7 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-5 dispatch (-7)', j['value'], j['ms_per_step'], j['telemetry'])" >> gpurun_out/${T}_ab_de.txt
  timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DE dispatch (default)', j['value'], j['ms_per_step'], j['telemetry'])" >> gpurun_out/${T}_ab_de.txt
done
cat gpurun_out/${T}_ab_de.txt
timeout 1500 python scratch/energy_ledger.py --steps 30 --out gpurun_out/${T}_energy_ledger.txt > gpurun_out/${T}_energy_ledger.log 2>&1
cat gpurun_out/${T}_energy_ledger.log | tail -60
