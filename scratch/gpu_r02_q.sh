cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "--batch 4 --streams 3" "--batch 4 --streams 2" "--batch 4 --streams 4" "--batch 6 --streams 3" "--batch 8 --streams 2" "--batch 6 --streams 2" "--batch 3 --streams 3"; do
  echo "== $cfg: $(timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-f32-variant --profile-steps 0 $cfg 2>&1 | tail -1 | grep -o '"value": [0-9.]*, "ms_per_step": [0-9.]*')"
done
