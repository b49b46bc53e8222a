# round 5, call W: the training step with more of its single-image pointwise convolutions on frcnn_gemm_h2 (cfg.HIP.H2_MIN_TILES: 150 keeps
# block3 conv1 -- 38 tiles at one image -- on the f32 MFMA with split-K + finishing pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_w}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_c5_h2_min_tiles.txt
: > $OUT
for rep in 1 2; do
for mt in ${MTS:-150 76 38 16}; do
export LABEL="H2_MIN_TILES=$mt"
timeout 200 python bench.py --config c5 --steps 20 --warmup 5 --hip H2_MIN_TILES=$mt 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('c5', os.environ['LABEL'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['roofline'].get('pipes'))" >> $OUT
done
done
cut -c1-330 $OUT
