# round 5, call S: is the batch-1 latency host-bound on the boxes that give 4.95 ms?  host enqueue time per graph launch beside the step time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_s}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_latency_host_share.txt
: > $OUT
nproc >> $OUT; uptime >> $OUT
for rep in 1 2 3; do
timeout 120 python bench.py --config c2 --batch 1 --streams 1 --steps 100 --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency ms/step', d['ms_per_step'], 'host enqueue ms/step', d['config'].get('host_enqueue_ms_per_step'), d.get('telemetry'))" >> $OUT
done
timeout 120 python bench.py --steps 30 --warmup 6 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('throughput', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue ms/step', d['config'].get('host_enqueue_ms_per_step'), d.get('telemetry'))" >> $OUT
cat $OUT
