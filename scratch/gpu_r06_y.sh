# round 6, final sanity after the last rebuild: smoke, the Winograd / dense tests, a short bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_h2_gpu.py -q 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-f32-variant --steps 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['traffic_source'], j['summary'])"
