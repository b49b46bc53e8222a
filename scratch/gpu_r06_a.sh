# round 6, call A: the whole GPU suite (new: streaming shapes, crop backward plans, masked light-boundary twins, mangled _nms, derived-set
# generation) + the default bench line on this round's first box (the baseline every later A/B of the round is read against)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -x -k "streaming or crop_and_resize_bwd or masked or mangled or filter_images_added" > gpurun_out/${TAG:-r06_a}_new_tests.txt 2>&1
tail -30 gpurun_out/${TAG:-r06_a}_new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/${TAG:-r06_a}_gpu_tests.txt 2>&1
tail -30 gpurun_out/${TAG:-r06_a}_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/${TAG:-r06_a}_bench.json 2> gpurun_out/${TAG:-r06_a}_bench.err
tail -c 1500 gpurun_out/${TAG:-r06_a}_bench.json
