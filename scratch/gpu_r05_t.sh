# round 5, call T: batch-1 latency with the 3x3 layers of block1 / block2 / the RPN on the direct kernel (one launch instead of three per layer)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_t}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_latency_direct_scopes.txt
: > $OUT
BENCH=${BENCH:-python bench.py --config c2 --batch 1 --streams 1 --steps 100 --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs}
for rep in 1 2; do
for sc in "()" "('block1',)" "('block1','block2')" "('block2',)" "('rpn_conv',)"; do
export LABEL="$sc"
timeout 120 $BENCH --hip "WINOGRAD_DIRECT_SCOPES=$sc" 2>/dev/null | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('latency direct scopes', os.environ['LABEL'], d['ms_per_step'], (d.get('telemetry') or {}).get('other_cards_max_w'))" >> $OUT
done
done
cat $OUT
