# round 3, call a: the whole GPU suite at HEAD (new full-size cases c1 / c4 / c5 included) + the parity printout
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r03_a
mkdir -p gpurun_out
rm -f gpurun_out/fullsize_parity.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${TAG}_pytest.log
cp gpurun_out/fullsize_parity.txt gpurun_out/${TAG}_fullsize_parity.txt 2>/dev/null
tail -15 gpurun_out/${TAG}_pytest.log
cut -c1-700 gpurun_out/${TAG}_fullsize_parity.txt | tail -40
