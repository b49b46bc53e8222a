# round 5, call I: replay tests + the tests that failed in call H2 + c5 A/B: eager / replayed / replayed with the stream picker, also under the data-parallel rules
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_i}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_replay_gpu.py tests/test_train_dp_gpu.py "tests/test_train_gpu.py::test_sgd_steps_on_a_fixed_batch_reduce_the_loss" "tests/test_train_gpu.py::test_wgrad_side_streams_change_nothing" "tests/test_train_gpu.py::test_snapshot_and_resume_continue_the_same_run" -m gpu -q --timeout=300 --maxfail=4 2>&1 | tail -60) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log | tail -45
OUT=gpurun_out/${TAG}_c5_ab_replay.txt
: > $OUT
run() {
  timeout 240 python bench.py --config c5 --steps 20 --warmup 5 "$@" 2>>gpurun_out/${TAG}_c5.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('c5 [$*]', d['value'], d['ms_per_step'], 'host_enqueue', c.get('host_enqueue_ms_per_step'), c.get('launch'), c.get('stream_pick'))
except Exception as e:
    print('c5 [$*] FAILED', e)" >> $OUT
}
run
run --no-train-replay
run --pick-streams 6
run
run --no-train-replay
run --dp-constrained
run --dp-constrained --no-train-replay
run --dp-constrained --pick-streams 6
cat $OUT; tail -5 gpurun_out/${TAG}_c5.err
