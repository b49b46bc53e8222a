cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { echo "== $*" >> gpurun_out/sweep.log; timeout 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --profile-steps 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep.log 2>&1; }
run
run --batch 4 --streams 2
run --batch 6 --streams 2
run --batch 6 --streams 3
run --batch 8 --streams 2
run --batch 3 --streams 4
run --batch 2 --streams 4
run --batch 5 --streams 3
run --crop-slabs 1
run --crop-slabs 4
run --crop-slabs 16
run --no-fused-mean
run --no-overlap
run --batch 1 --streams 1 --no-overlap
run --winograd-f2 ""
run --batch 1 --streams 1
run --batch 1 --streams 3
run --batch 2 --streams 1
cat gpurun_out/sweep.log
