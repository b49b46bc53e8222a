# round 6, call C: (1) the batched-forward failure of call B under three settings; (2) the deferred epilogue: bit equality + time per shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${TAG:-r06_c}
timeout 300 python -m pytest tests/test_network_gpu.py -q -x -k "batched_forward" > gpurun_out/${T}_batched_alone.txt 2>&1; tail -5 gpurun_out/${T}_batched_alone.txt
FRCNN_SCOPE_CAP=64 timeout 600 python -m pytest tests/test_network_gpu.py -q -x > gpurun_out/${T}_network_cap64.txt 2>&1; tail -5 gpurun_out/${T}_network_cap64.txt
FRCNN_SCOPE_POISON=1 timeout 600 python -m pytest tests/test_network_gpu.py -q -x > gpurun_out/${T}_network_poison.txt 2>&1; tail -5 gpurun_out/${T}_network_poison.txt
timeout 900 python scratch/h2_conv3.py 9,31,40,33,41,21 b4c3x8p,b4c3x8,b3c3x8p,b3c3x8,b2c3x8,w7x8,w3x8,b3scx8,b4c1x8,b3c1x8,b4c3x1,b3c3x1 > gpurun_out/${T}_h2_de.txt 2>&1
cat gpurun_out/${T}_h2_de.txt
