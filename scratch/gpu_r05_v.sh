# round 5, call V: rocprofv3 counters of the SHIPPED conv3-class launch alone (block4 conv3 as it runs since the trunk is planes: residual read as
# planes, planes only out, cfg 31) -> profiles/r05_counters_conv3.json, which bench.py copies into roofline.counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_v}
mkdir -p gpurun_out/pmc_v
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES TA_BUSY_avr"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_v/$tag -o p -- python $GRAFT_REPO_ROOT/scratch/h2_conv3.py 31 b4c3x8p --single 6 > $GRAFT_REPO_ROOT/gpurun_out/pmc_v/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_kernel.py gpurun_out/pmc_v k_gemm_h2 gpurun_out/${TAG}_counters_conv3.json "profiles/${TAG}_pmc_b4c3x8p_cfg31.txt (block4 conv3 alone as shipped: residual as planes, planes out, cfg 31; 8 images)" > gpurun_out/${TAG}_pmc_b4c3x8p_cfg31.txt 2>&1
cat gpurun_out/${TAG}_pmc_b4c3x8p_cfg31.txt; cat gpurun_out/${TAG}_counters_conv3.json | head -40
rm -rf gpurun_out/pmc_v
