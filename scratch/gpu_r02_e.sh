cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { echo "== $*" >> gpurun_out/stagger.log; timeout 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --profile-steps 2 --layer-report gpurun_out/layers_$1.txt "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])" >> gpurun_out/stagger.log 2>&1; }
run base
run s4 --stagger 4
run s2 --stagger 2
run s20 --stagger 20
run s16 --stagger 16
run base2
run lat_base --batch 1 --streams 1
run lat_s4 --batch 1 --streams 1 --stagger 4
cat gpurun_out/stagger.log
grep "block4/unit_2" gpurun_out/layers_base.txt gpurun_out/layers_s4.txt gpurun_out/layers_s2.txt | cut -c1-150
