# round 5, call E: the new by-shape dispatch (31 / 33 / 21) against round 4's (-2: 9 / 12 / 21) in the 3-chain pipeline, interleaved A/B/A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_e}_ab_dispatch.txt
: > $OUT
for rep in 1 2; do
  for c in ${CFGS:--2 -1}; do
    timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --h2-cfg $c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h2-cfg $c', d['value'], d['ms_per_step'], d.get('telemetry'))" >> $OUT
  done
done
cat $OUT
