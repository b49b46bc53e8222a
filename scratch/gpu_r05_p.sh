# round 5, call P: images per chain x chains in flight, re-measured at the round-5 state (round 4: profiles/r04_n_batch_chains_sweep.txt)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_p}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_batch_chains_sweep.txt
: > $OUT
for bs in "8 3" "8 2" "8 4" "6 4" "10 3" "12 2" "12 3" "16 2" "8 3"; do
  set -- $bs
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --batch $1 --streams $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d.get('telemetry') or {}; print('batch $1 chains $2', d['value'], d['ms_per_step'], t.get('sclk_mhz'), t.get('socket_w'), t.get('other_cards_max_w'))" >> $OUT
done
cat $OUT
