cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_network_gpu.py -m gpu -q -k batched_forward 2>&1 | grep -v "^  \|Warning" | head -80) > gpurun_out/pytest_one.log
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/pytest.log
cat gpurun_out/pytest_one.log | cut -c1-400 | head -60; tail -4 gpurun_out/pytest.log
