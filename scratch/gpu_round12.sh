cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# library sgemm reference point for the block4 1x1 GEMMs
python - <<'PY' 2>&1 | tail -8
import torch, numpy as np
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for (M, N, K) in ((58800, 512, 2048), (58800, 2048, 512), (9576, 256, 1024), (9576, 1024, 256)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    for name, fn in (("matmul(a, b.T)", lambda: torch.matmul(a, b.t())),):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 8 * 1e3)
        us = float(np.median(ts))
        print("torch f32 %s M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
PY
mkdir -p gpurun_out/pmc3
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc3/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_traffic.py gpurun_out/pmc3 gpurun_out/r01_e_pmc_traffic.json
find gpurun_out/pmc3 -name "*.csv" -size +8M -delete
