# round 6, call D: the deferred epilogue after the register diet (bit equality + time per shape), the network / streaming tests after the
# image-address fix
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${TAG:-r06_d}
timeout 900 python scratch/h2_conv3.py 9,31,40,33,41,21 b4c3x8p,b4c3x8,b3c3x8p,b3c3x8,b2c3x8,w7x8,w3x8,b3scx8,b4c1x8,b3c1x8 > gpurun_out/${T}_h2_de.txt 2>&1
cat gpurun_out/${T}_h2_de.txt
timeout 900 python -m pytest tests/test_network_gpu.py tests/test_streaming_shapes_gpu.py -q > gpurun_out/${T}_network.txt 2>&1; tail -5 gpurun_out/${T}_network.txt
