# PMC look at k_gemm_h2 alone (b4c1x4, cfg 0): where do the wave cycles go?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_i
mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > gpurun_out/${TAG}_sq_counters.txt
wc -l gpurun_out/${TAG}_sq_counters.txt
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f2)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -o p -- python $GRAFT_REPO_ROOT/scratch/h2_sweep.py 0 b4c1x4,b3c3x4 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log 2>&1
  tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_kernel.py gpurun_out/pmc "k_gemm_h2<128, 128, 64, 64, 2, 2, 0>" > gpurun_out/${TAG}_pmc_gemm_h2.txt 2>&1
cat gpurun_out/${TAG}_pmc_gemm_h2.txt
rm -rf gpurun_out/pmc
