# round 5, call X: cfg.HIP.H2_TRAIN_MIN_TILES = 320 as the TRAIN-mode default: the training tests + the c5 lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_x}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_replay_gpu.py tests/test_train_dp_gpu.py tests/test_chain_fusion_gpu.py "tests/test_fullsize_gpu.py::test_fullsize_train_step_parity" -m gpu -q --timeout=400 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
OUT=gpurun_out/${TAG}_c5_train_min_tiles.txt
: > $OUT
for rep in 1 2; do
for a in "H2_TRAIN_MIN_TILES=150" "H2_TRAIN_MIN_TILES=320"; do
export LABEL="$a"
timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --hip $a 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('c5', os.environ['LABEL'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" >> $OUT
export LABEL="$a --dp-constrained"
timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --hip $a --dp-constrained 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('c5', os.environ['LABEL'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))" >> $OUT
done
done
cat $OUT
