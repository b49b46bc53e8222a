#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab; mkdir -p $O
timeout 300 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline > $O/c5_side.json 2> $O/c5_side.err
timeout 300 python bench.py --config c5 --steps 30 --warmup 5 --no-cpu-baseline --no-wgrad-stream > $O/c5_main.json 2> $O/c5_main.err
cat $O/c5_side.json $O/c5_main.json | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_c5 -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/overlap.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"] if "Stream_Id" in r else r.get("Queue_Id"), r["Kernel_Name"][:40]) for r in rows))
# last 40% of the trace = steady steps
t0 = ev[int(len(ev) * 0.6)][0]
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
by = collections.defaultdict(int)
for s, e, q, n in ev:
    by[q] += e - s
# union busy time
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, q, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels", len(ev), "span ms", span / 1e6, "union busy ms", busy / 1e6, "idle frac", 1 - busy / span)
for q, t in sorted(by.items(), key=lambda x: -x[1]):
    print("queue/stream", q, "sum ms", t / 1e6, "frac of span", t / span)
print(list(rows[0].keys()))
PY
cat $O/overlap.txt
