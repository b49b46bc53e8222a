# round 5, call O: the crop as one workgroup per (roi, slab): isolated timings + bit equality, the crop tests, pipeline A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_o}
mkdir -p gpurun_out
(timeout 200 python scratch/crop_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/${TAG}_crop_forms.txt
cat gpurun_out/${TAG}_crop_forms.txt
(timeout 400 python -m pytest tests/test_detect_gpu.py tests/test_edges_gpu.py tests/test_boundary_gpu.py -m gpu -q -k "crop" --timeout=300 2>&1 | tail -5) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
OUT=gpurun_out/${TAG}_ab_crop_form.txt
: > $OUT
for rep in 1 2 3; do
  for c in 1 0; do
    timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-f32-variant --no-other-configs --profile-steps 0 --tune d:5=$c 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('crop form $c', d['value'], d['ms_per_step'], d.get('telemetry'))" >> $OUT
  done
done
cat $OUT
