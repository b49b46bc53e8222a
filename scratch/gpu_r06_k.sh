# round 6, call K: the training run over changing image shapes (bounded scopes, reproducible bits)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_streaming_shapes_gpu.py -q > gpurun_out/r06_k_streaming.txt 2>&1; tail -15 gpurun_out/r06_k_streaming.txt
