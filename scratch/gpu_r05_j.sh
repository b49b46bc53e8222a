# round 5, call J: replay + DP tests; why --dp-constrained dies; c5 under the data-parallel rules: eager / replayed / replayed + stream picker
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_j}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_replay_gpu.py tests/test_train_dp_gpu.py -m gpu -q --timeout=400 2>&1 | tail -60) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log | tail -45
timeout 200 python -X faulthandler bench.py --config c5 --steps 3 --warmup 3 --dp-constrained > gpurun_out/${TAG}_dp_debug.txt 2>&1; echo "dp debug rc=$?" >> gpurun_out/${TAG}_dp_debug.txt
grep -v "amdgpu.ids" gpurun_out/${TAG}_dp_debug.txt | tail -40 | cut -c1-600
OUT=gpurun_out/${TAG}_c5_dp_ab.txt
: > $OUT
run() {
  timeout 300 python bench.py --config c5 --steps 20 --warmup 5 "$@" 2>>gpurun_out/${TAG}_c5.err | grep '^{' | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('c5 [$*]', d['value'], d['ms_per_step'], 'host_enqueue', c.get('host_enqueue_ms_per_step'), c.get('launch'), c.get('stream_pick'))
except Exception as e:
    print('c5 [$*] FAILED', e)" >> $OUT
}
run --dp-constrained
run --dp-constrained --no-train-replay
run --dp-constrained --pick-streams 6
run --dp-constrained
run --dp-constrained --no-train-replay
run --dp-constrained --pick-streams 6
run
cat $OUT
