cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_y
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_dp_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "train" 2>&1 | tail -4) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
grep -A4 "TRAIN" gpurun_out/fullsize_parity.txt | cut -c1-420 > gpurun_out/${TAG}_train_parity.txt; cat gpurun_out/${TAG}_train_parity.txt
mkdir -p gpurun_out/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/run.log 2>&1 )
DB=$(find gpurun_out/prof -name '*.db' | head -1); python scratch/rocpd_by_shape.py $DB gpurun_out/${TAG}_train_kernels_by_shape.txt 60 > /dev/null; find gpurun_out/prof -name '*.db' -delete; rm -rf gpurun_out/prof
head -30 gpurun_out/${TAG}_train_kernels_by_shape.txt | cut -c1-140
for h in 150 100000; do
timeout 300 python bench.py --config c5 --steps 16 --warmup 4 --no-cpu-baseline --h2-min-tiles $h 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 H2_MIN_TILES $h:', d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"
done > gpurun_out/${TAG}_c5.txt; cat gpurun_out/${TAG}_c5.txt
