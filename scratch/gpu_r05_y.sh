# round 5, call Y: cfg.HIP.H2_TRAIN_MIN_TILES = 320 as the TRAIN-mode default: the full-size training tests + one c5 line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_y}
mkdir -p gpurun_out
(timeout 230 python -m pytest "tests/test_fullsize_gpu.py::test_fullsize_train_step_parity[shipped]" "tests/test_replay_gpu.py::test_fullsize_c5_step_is_deterministic_and_replay_equals_eager" tests/test_train_dp_gpu.py "tests/test_replay_gpu.py::test_the_stream_picker_finishes_and_changes_no_bit" -m gpu -q -x --timeout=200 2>&1 | tail -30) > gpurun_out/${TAG}_pytest.log
tail -12 gpurun_out/${TAG}_pytest.log | cut -c1-220
timeout 100 python bench.py --config c5 --steps 10 --warmup 4 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 default', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['roofline']['pipes']['h2']['share_of_launched_flops'])" > gpurun_out/${TAG}_c5.txt 2>&1
cat gpurun_out/${TAG}_c5.txt
