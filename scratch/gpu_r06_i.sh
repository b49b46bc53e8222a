# round 6, call I: the whole GPU suite at this commit; the training step (c5) with the masked light-boundary twins, TRAIN tile threshold 320 / 150
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${TAG:-r06_i}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/${T}_gpu_tests.txt 2>&1; tail -8 gpurun_out/${T}_gpu_tests.txt
for th in 320 150 320 150; do
  timeout 400 python bench.py --config c5 --steps 20 --hip H2_TRAIN_MIN_TILES=$th 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 H2_TRAIN_MIN_TILES=$th', j['value'], j['ms_per_step'], j['config'].get('host_enqueue_ms_per_step'), j['roofline'].get('frac'))" >> gpurun_out/${T}_c5.txt
done
cat gpurun_out/${T}_c5.txt
