# clocks / power while the bench runs (is the pipeline clock-limited?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocm-smi --showclocks --showpower 2>&1 | head -40 > gpurun_out/i_smi_idle.txt
for S in 3 1; do
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/i_smi_s$S.txt &
  SMI=$!
  timeout 300 python bench.py --steps 600 --warmup 5 --no-cpu-baseline --streams $S --profile-steps 0 2>&1 | tail -1 | cut -c1-420 > gpurun_out/i_bench_s$S.json
  wait $SMI
done
cat gpurun_out/i_smi_idle.txt | head -30; for S in 3 1; do echo "== streams $S"; cut -c150-330 gpurun_out/i_bench_s$S.json; sed -n '20,50p' gpurun_out/i_smi_s$S.txt | cut -c1-250; done
