# round 5, call F: trunk as planes only (cfg.HIP.H2_TRUNK_PLANES) with the round-5 epilogue, in the 3-chain pipeline, interleaved
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_f}_ab_trunk_planes.txt
: > $OUT
for rep in 1 2; do
  for c in 0 1; do
    timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --h2-trunk-planes $c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('trunk-planes $c', d['value'], d['ms_per_step'], d.get('telemetry'))" >> $OUT
  done
done
cat $OUT
