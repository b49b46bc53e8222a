cd $GRAFT_REPO_ROOT
for bs in "8 3" "12 3" "16 3" "16 2" "12 2"; do set -- $bs
  echo -n "batch $1 streams $2: "; timeout 250 python bench.py --no-cpu-baseline --profile-steps 0 --batch $1 --streams $2 --steps 18 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/sweep_h2.txt
