# round 5, call K: relaxed data-parallel rules (two filter-gradient streams behind an ordering stream, solver behind each reduced bucket) + stream picker by default
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_k}
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_replay_gpu.py tests/test_train_dp_gpu.py tests/test_train_gpu.py tests/test_network_gpu.py "tests/test_fullsize_gpu.py::test_fullsize_batch_invariance_under_the_shipped_configuration" -k "replay or repeats or picker or recording or fullsize_c5 or replicas or sgd_steps or side_streams or snapshot_and_resume or extract_head or fused_tail_mean or batch_invariance" -m gpu -q --timeout=400 2>&1 | tail -40) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log | tail -30
OUT=gpurun_out/${TAG}_c5_dp_relaxed.txt
: > $OUT
run() {
  timeout 300 python bench.py --config c5 --steps 20 --warmup 5 "$@" 2>>gpurun_out/${TAG}_c5.err | grep '^{' | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('c5 [$*]', d['value'], d['ms_per_step'], 'host_enqueue', c.get('host_enqueue_ms_per_step'), c.get('launch'), c.get('stream_pick'))
except Exception as e:
    print('c5 [$*] FAILED', e)" >> $OUT
}
run
run --dp-constrained
run
run --dp-constrained
run --dp-constrained --pick-streams 0
cat $OUT | cut -c1-900; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/${TAG}_c5.err | tail -5
