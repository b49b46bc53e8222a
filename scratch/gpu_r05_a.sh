# round 5, call A: (1) what a store PATTERN costs (scratch/store_patterns.hip), (2) the conv3-class launches at the bench batch under the
# three shipped tile configurations (baseline for the epilogue work), (3) PMC passes on block4 conv3 ALONE (`bound`
# for this launch from counters), (4) the training step vs torch stream-pool position, with / without a one-rank RCCL group (reduced).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
./scratch/store_patterns > gpurun_out/r05_a_store_patterns.txt 2>&1
cat gpurun_out/r05_a_store_patterns.txt
timeout 300 python scratch/h2_conv3.py 9,21,12 > gpurun_out/r05_a_h2_conv3_baseline.txt 2>&1
cat gpurun_out/r05_a_h2_conv3_baseline.txt
mkdir -p gpurun_out/pmc_a
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES TA_BUSY_avr"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a/$tag -o p -- python $GRAFT_REPO_ROOT/scratch/h2_conv3.py 9 b4c3x8 --single 6 > $GRAFT_REPO_ROOT/gpurun_out/pmc_a/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_kernel.py gpurun_out/pmc_a k_gemm_h2 > gpurun_out/r05_a_pmc_b4c3x8_cfg9.txt 2>&1
cat gpurun_out/r05_a_pmc_b4c3x8_cfg9.txt
rm -rf gpurun_out/pmc_a
timeout 400 python scratch/stream_pool_sweep.py gpurun_out/r05_a_stream_pool_sweep.txt 0,1,2,3,4,6,8
cat gpurun_out/r05_a_stream_pool_sweep.txt
