# round 5, call L: bit identity of cfgs 30-34 / the by-shape rules on the conv3 class incl. the residual as planes; isolated timings of the
# shipped (trunk-as-planes) conv3 forms; pipeline A/B of the by-shape rules -1 (shipped) / -3 (ping-pong with 16-byte plane stores) /
# -4 (... and the conv3 class on the ping-pong schedule), interleaved
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_l}
mkdir -p gpurun_out
(timeout 400 python -m pytest tests/test_h2_gpu.py -m gpu -q -k "ping_pong" --timeout=300 2>&1 | tail -8) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
(timeout 300 python scratch/h2_conv3.py 9,31,21,34 b4c3x8p,b3c3x8p,b4c3x8,b4c1x8,b4c3x8m 2>&1 | grep -v amdgpu.ids) > gpurun_out/${TAG}_h2_conv3_trunk_planes.txt
cat gpurun_out/${TAG}_h2_conv3_trunk_planes.txt
OUT=gpurun_out/${TAG}_ab_dispatch_rules.txt
: > $OUT
for rep in 1 2; do
  for c in -1 -3 -4; do
    timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --hip H2_TILE_CFG=$c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('H2_TILE_CFG $c', d['value'], d['ms_per_step'], d.get('telemetry'))" >> $OUT
  done
done
cat $OUT
