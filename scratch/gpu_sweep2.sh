cd $GRAFT_REPO_ROOT
for bs in "4 3" "6 3" "8 3" "4 4" "6 2" "8 2" "5 3"; do set -- $bs
  echo -n "batch $1 streams $2: "; timeout 200 python bench.py --no-cpu-baseline --profile-steps 0 --batch $1 --streams $2 --steps 24 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/sweep_h.txt
