cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for M in x3 f32; do
  ( for i in $(seq 1 44); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/o_smi_$M.txt &
  SMI=$!
  timeout 300 python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --mfma $M 2>&1 | tail -1 | grep -o '"value": [0-9.]*, "ms_per_step": [0-9.]*' > gpurun_out/o_bench_$M.txt
  wait $SMI
  echo "== $M $(cat gpurun_out/o_bench_$M.txt)"
  echo "sclk: $(grep -o 'sclk clock level: [0-9S]: ([0-9]*Mhz)' gpurun_out/o_smi_$M.txt | grep -o '([0-9]*' | tr -d '(' | awk '$1>1000' | sort -n | awk '{a[NR]=$1} END{print "min",a[1],"median",a[int((NR+1)/2)],"max",a[NR],"n",NR}')"
  echo "power: $(grep -o 'Power (W): [0-9.]*' gpurun_out/o_smi_$M.txt | awk '{print $3}' | awk '$1>900' | sort -n | awk '{a[NR]=$1} END{print "min",a[1],"median",a[int((NR+1)/2)],"max",a[NR],"n",NR}')"
done
