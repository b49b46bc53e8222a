# round 5, call B: the light tile boundary (cfg 30), 16-byte plane stores (cfg 32), both (cfg 31), 64-row form (cfg 33) against cfg 9 / 21 / 12
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scratch/h2_conv3.py ${CFGS:-9,30,31,32,33,21,12} ${SHAPES:-b4c3x8,b4c3x8m,b3c3x8,b2c3x8,w7x8,w3x8,b3scx8,b4c3x1,b3c3x1} > gpurun_out/${TAG:-r05_b}_h2_conv3.txt 2>&1
cat gpurun_out/${TAG:-r05_b}_h2_conv3.txt
