# round 5, call H: the training-step tests (masked h2 dispatch fix, deterministic crop backward, recorded / replayed step) + c5 A/B replay vs eager
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_h}
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_replay_gpu.py tests/test_chain_fusion_gpu.py tests/test_train_gpu.py tests/test_train_dp_gpu.py "tests/test_fullsize_gpu.py::test_fullsize_train_step_parity" -m gpu -q --timeout=300 2>&1 | tail -60) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log | tail -40
OUT=gpurun_out/${TAG}_c5_ab_replay.txt
: > $OUT
for rep in 1 2; do
  for mode in "" "--no-train-replay"; do
    timeout 200 python bench.py --config c5 --steps 20 --warmup 5 $mode 2>gpurun_out/${TAG}_c5.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('c5 [$mode]', d['value'], d['ms_per_step'], 'host_enqueue', c.get('host_enqueue_ms_per_step'), c.get('launch'))" >> $OUT
  done
done
for mode in "" "--no-train-replay"; do
  timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --dp-constrained $mode 2>>gpurun_out/${TAG}_c5.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('c5 dp-constrained [$mode]', d['value'], d['ms_per_step'], 'host_enqueue', c.get('host_enqueue_ms_per_step'), c.get('launch'))" >> $OUT
done
cat $OUT; tail -5 gpurun_out/${TAG}_c5.err
