# round 5, call R: runtime switches of the HIP runtime that touch kernel-dispatch latency, on the two launch-bound lines (batch-1 latency = one
# hipGraph of ~300 kernel nodes per image; the C5 training step = ~1 300 launches over 5 streams)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_r}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_runtime_switches.txt
: > $OUT
lat() { env "$@" timeout 120 python bench.py --config c2 --batch 1 --streams 1 --steps 100 --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency [$*]', d['ms_per_step'])" >> $OUT; }
c5() { env "$@" timeout 200 python bench.py --config c5 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 [$*]', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" >> $OUT; }
for rep in 1 2; do
  lat A=0
  lat HIP_FORCE_DEV_KERNARG=1
  lat HIP_FORCE_DEV_KERNARG=0
  lat DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  lat DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  lat GPU_MAX_HW_QUEUES=2
done
c5 A=0
c5 HIP_FORCE_DEV_KERNARG=1
c5 HIP_FORCE_DEV_KERNARG=0
c5 A=0
cat $OUT
