# round 3: shape-based tile dispatch, residual trunk as planes only (cfg.HIP.H2_TRUNK_PLANES) A/B, min-tiles sweep for single images
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_h
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 900 python -m pytest tests/test_h2_gpu.py tests/test_network_gpu.py -m gpu -q -x -s 2>&1 | grep -E "h2 launches|h2 path|passed|failed|Error|error|assert|fused tail" | tail -40) > gpurun_out/${TAG}_tests_a.log
cat gpurun_out/${TAG}_tests_a.log
(timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -k "(shipped-c or shipped_h2) and not train" 2>&1 | tail -5) > gpurun_out/${TAG}_tests_fullsize.log
cat gpurun_out/${TAG}_tests_fullsize.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt
cut -c1-400 gpurun_out/${TAG}_fullsize_parity.txt
for tp in 1 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --h2-trunk-planes $tp 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('trunk planes $tp:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done > gpurun_out/${TAG}_ab.txt 2>&1
for mt in 150 64 24 1; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --batch 1 --streams 1 --h2-min-tiles $mt 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch 1, 1 chain, H2_MIN_TILES $mt:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done >> gpurun_out/${TAG}_ab.txt 2>&1
for mt in 64 24; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-variant --profile-steps 0 --h2-min-tiles $mt 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch 4 x 3 chains, H2_MIN_TILES $mt:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done >> gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
