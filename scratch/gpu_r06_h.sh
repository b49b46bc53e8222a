# round 6, call H: cache-policy experiments on the shipped conv3 launch: `nt` on the once-touched streams (cfg 42), the 8 x 8 panel order (43), both (44)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${TAG:-r06_h}
mkdir -p gpurun_out/pmc_h
timeout 600 python scratch/h2_conv3.py 9,31,40,42,43,44 b4c3x8p,b3c3x8p > gpurun_out/${T}_h2_cache_policy.txt 2>&1; cat gpurun_out/${T}_h2_cache_policy.txt
cd /tmp
for c in 40 42 43 44; do
  timeout 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ TCC_REQ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_h/c$c -o p -- python $GRAFT_REPO_ROOT/scratch/h2_conv3.py $c b4c3x8p --single 6 > $GRAFT_REPO_ROOT/gpurun_out/pmc_h/c$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
for c in 40 42 43 44; do echo "== cfg $c (block4 conv3, residual as planes, planes out, 8 images)"; python scratch/pmc_kernel.py gpurun_out/pmc_h/c$c k_gemm_h2; done > gpurun_out/${T}_pmc_cache_policy.txt 2>&1
cat gpurun_out/${T}_pmc_cache_policy.txt
rm -rf gpurun_out/pmc_h
