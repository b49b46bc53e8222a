cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python scratch/run_configs.py 2>&1 | grep -E "^C[1-5]" | tee gpurun_out/all_configs_e.txt
for bs in "4 3" "8 2" "8 3" "4 4" "2 4" "6 3"; do set -- $bs
  echo -n "batch $1 streams $2: "; timeout 200 python bench.py --no-cpu-baseline --profile-steps 0 --batch $1 --streams $2 --steps 24 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/sweep_e.txt
timeout 300 python scratch/wino_err.py 2>&1 | tail -6 > gpurun_out/wino_err.txt
