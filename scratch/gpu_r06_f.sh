# round 6, call F: Winograd transforms (row-per-thread threshold), counters of the shipped conv3 launch (cfg 40) + FETCH_SIZE calibration on
# 8-byte / 16-byte residual loads, counters of the block3 transforms
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${TAG:-r06_f}
mkdir -p gpurun_out/pmc_f
timeout 600 python scratch/wino_bench.py 256,512,100000000 > gpurun_out/${T}_wino_bench.txt 2>&1; cat gpurun_out/${T}_wino_bench.txt
cd /tmp
pmc() {   # name, counter set, command...
  local name=$1; shift; local set=$1; shift
  local tag=$(echo $set | cut -d' ' -f1)
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_f/$name/$tag -o p -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_f/$name.$tag.log 2>&1
}
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ TCC_EA0_RDREQ_32B"; do
  pmc conv3_cfg40 "$set" python $GRAFT_REPO_ROOT/scratch/h2_conv3.py 40 b4c3x8p --single 6
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  pmc cal_cfg31 "$set" python $GRAFT_REPO_ROOT/scratch/h2_conv3.py 31 cal_k128 --single 6
  pmc cal_cfg40 "$set" python $GRAFT_REPO_ROOT/scratch/h2_conv3.py 40 cal_k128 --single 6
  pmc wino_in_b3 "$set" python $GRAFT_REPO_ROOT/scratch/wino_bench.py "single=block3 conv2 x8:input:6"
  pmc wino_out_b3 "$set" python $GRAFT_REPO_ROOT/scratch/wino_bench.py "single=block3 conv2 x8:output:6"
  pmc wino_in_7 "$set" python $GRAFT_REPO_ROOT/scratch/wino_bench.py "single=tail conv2 7x7 x8:input:6"
  pmc wino_out_7 "$set" python $GRAFT_REPO_ROOT/scratch/wino_bench.py "single=tail conv2 7x7 x8:output:6"
done
cd $GRAFT_REPO_ROOT
python scratch/pmc_kernel.py gpurun_out/pmc_f/conv3_cfg40 k_gemm_h2 gpurun_out/${T}_counters_conv3.json "profiles/${T}_pmc_b4c3x8p_cfg40.txt (block4 conv3 alone as shipped in round 6: residual as planes, planes out, cfg 40 = deferred epilogue; 8 images)" > gpurun_out/${T}_pmc_b4c3x8p_cfg40.txt 2>&1
cat gpurun_out/${T}_pmc_b4c3x8p_cfg40.txt
for n in cal_cfg31 cal_cfg40; do echo "== $n (X planes 60.2 MB by 16-byte direct-to-LDS loads + residual planes 60.2 MB + filter 64 KB read; 60.2 MB written)"; python scratch/pmc_kernel.py gpurun_out/pmc_f/$n k_gemm_h2; done > gpurun_out/${T}_pmc_calibration.txt 2>&1
for n in wino_in_b3 wino_out_b3 wino_in_7 wino_out_7; do echo "== $n"; grep algorithmic gpurun_out/pmc_f/$n.FETCH_SIZE.log; python scratch/pmc_kernel.py gpurun_out/pmc_f/$n k_wino; done > gpurun_out/${T}_pmc_wino.txt 2>&1
cat gpurun_out/${T}_pmc_calibration.txt gpurun_out/${T}_pmc_wino.txt
rm -rf gpurun_out/pmc_f
