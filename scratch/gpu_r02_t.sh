cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in -1 4 -1 4; do
  echo "== --x3-config $c: $(timeout 200 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-f32-variant --profile-steps 0 --x3-config $c 2>&1 | tail -1 | grep -o '"value": [0-9.]*, "ms_per_step": [0-9.]*')"
done
