# round 5, call M: by-shape rules -1 / -4 / -5 / -6 in the pipeline, interleaved x 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_m}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_dispatch_rules.txt
: > $OUT
for rep in 1 2 3; do
  for c in ${CFGS:--1 -4 -5 -6}; do
    timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --hip H2_TILE_CFG=$c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('H2_TILE_CFG $c', d['value'], d['ms_per_step'], d.get('telemetry'))" >> $OUT
  done
done
cat $OUT
