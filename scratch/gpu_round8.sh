cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_network_gpu.py -m gpu -q --timeout 300 --tb=line 2>&1 | grep -v "^$" | cut -c1-300 | tail -5
for cfg in "4 2" "4 3" "6 2" "8 2"; do set -- $cfg; echo "batch $1 streams $2"; timeout 300 python bench.py --steps $((48 / $1)) --warmup 4 --no-cpu-baseline --batch $1 --streams $2 --profile-steps 0 2>&1 | tail -1 | cut -c1-190; done
