cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01_h}
mkdir -p gpurun_out/pmc_$TAG
( cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --streams 1 --profile-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/$tag.log 2>&1
done )
python scratch/pmc_traffic.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc_traffic.json | head -12
find gpurun_out/pmc_$TAG -name "*.csv" -delete
cp gpurun_out/${TAG}_pmc_traffic.json profiles/r01_e_pmc_traffic.json      # the file bench.py reads (refreshed for this run only)
bash scratch/gpu_final.sh $TAG
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
