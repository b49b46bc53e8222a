# round 6, call B: the whole GPU suite after the per-shape scopes (call A stopped at 15 failures of one cause: the full-size harness read
# net._sess.buffers directly)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/${TAG:-r06_b}_gpu_tests.txt 2>&1
tail -40 gpurun_out/${TAG:-r06_b}_gpu_tests.txt
