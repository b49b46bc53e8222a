# round 5, call N: the batch-1 latency window by window (why does a warmed run give 4.96 ms where a 0.3 s run gave 4.02?), clocks beside it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05_n}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_latency_windows.txt
: > $OUT
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "GPU\[0\]" | grep -E "sclk|mclk|fclk|socclk|Power" | tr '\n' ' ' ; echo; sleep 0.5; done) > gpurun_out/${TAG}_smi.txt &
SMI=$!
for steps in 40 200; do
  echo "== steps $steps per window" >> $OUT
  FRCNN_BENCH_WINDOWS=14 timeout 200 python bench.py --config c2 --batch 1 --streams 1 --steps $steps --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs --warm-until-stable 2>&1 | grep -E "^window|^\{" | cut -c1-200 >> $OUT
done
echo "== eager launches (no hipGraph), steps 40" >> $OUT
FRCNN_BENCH_WINDOWS=8 timeout 200 python bench.py --config c2 --batch 1 --streams 1 --steps 40 --warmup 40 --profile-steps 0 --no-cpu-baseline --no-f32-variant --no-other-configs --warm-until-stable --no-graph 2>&1 | grep -E "^window|^\{" | cut -c1-200 >> $OUT
kill $SMI
cat $OUT | cut -c1-160
tail -40 gpurun_out/${TAG}_smi.txt | cut -c1-250
