#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ad; mkdir -p $O
for tn in 1 0; do for side in 2 0; do TN=$tn SIDE=$side python scratch/c5_phases.py 2>&1 | tail -4; done; done | tee $O/phases.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tn -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_tn -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200 > $O/kernel_stats_tn.txt
cat $O/kernel_stats_tn.txt | head -30
