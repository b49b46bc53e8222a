"""Which chain of the training step is the long one?  From a rocprofv3 (rocpd sqlite) kernel trace of `bench.py --config c5`:
the last full step (between two k_sgd_multi launches), per stream: launches, sum of durations, union busy time, idle gaps between
its consecutive kernels; whole GPU: time with >= 1 / exactly 1 / >= 2 kernels running; the main stream's timeline cut into phases
at marker kernels (softmax_ce = end of the forward pass, sgd = end of the sweep)."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
sid = "stream_id" if "stream_id" in cols else "queue_id"
rows = c.execute("select name, start, end, %s, grid_x, grid_y, grid_z from kernels order by start" % sid).fetchall()
# one step = from the start of one k_nms_reduce (once per step, in the proposal layer of the forward pass) to the start of the next
sgd = [i for i, r in enumerate(rows) if r[0].startswith("void k_nms_reduce") or r[0].startswith("k_nms_reduce")]
lo, hi = sgd[-2], sgd[-1]
step = rows[lo:hi]
t0, t1 = rows[sgd[-2]][1], rows[sgd[-1]][1]
f = open(out, "w")
f.write("# last full training step: %d launches, %.1f us from one k_nms_reduce to the next\n" % (len(step), (t1 - t0) / 1e3))
by = {}
for r in step: by.setdefault(r[3], []).append(r)
main = max(by, key=lambda s: len(by[s]))
f.write("# %-8s %8s %12s %12s %12s %10s\n" % ("stream", "launches", "sum_dur_us", "union_us", "gaps_us", "gaps>3us"))
for s, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    rs.sort(key=lambda r: r[1])
    tot = sum(r[2] - r[1] for r in rs); union = 0; gaps = 0; big = 0; end = None
    for r in rs:
        if end is None or r[1] >= end:
            if end is not None:
                gaps += r[1] - end; big += (r[1] - end) > 3000
            union += r[2] - r[1]; end = r[2]
        elif r[2] > end:
            union += r[2] - end; end = r[2]
    f.write("%-10s %8d %12.1f %12.1f %12.1f %10d%s\n" % (s, len(rs), tot / 1e3, union / 1e3, gaps / 1e3, big, "   <- main" if s == main else ""))
# whole-GPU concurrency histogram
ev = []
for r in step: ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
lvl = 0; last = ev[0][0]; hist = {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - last); last = t; lvl += d
f.write("# kernels running at once -> us of the step: " + ", ".join("%d: %.0f" % (k, v / 1e3) for k, v in sorted(hist.items())) + "\n")
# phases on the main stream
ms = sorted(by[main], key=lambda r: r[1])
# the step as cut above: [proposal layer .. losses] [reverse sweep, solver] [next forward pass up to its proposal layer]
marks = [i for i, r in enumerate(ms) if r[0].startswith("k_softmax_ce")]
fw_end = marks[-1] if marks else 0
sg = [i for i, r in enumerate(ms) if r[0].startswith("k_sgd_multi")]
bw_end = sg[-1] if sg else len(ms) - 1
def seg(name, a, b):
    rs = ms[a:b]
    if not rs: return
    dur = sum(r[2] - r[1] for r in rs)
    f.write("# main stream %-10s %5d launches, kernel time %9.1f us, span %9.1f us\n" % (name, len(rs), dur / 1e3, (rs[-1][2] - rs[0][1]) / 1e3))
seg("fwd tail", 0, fw_end + 1); seg("backward", fw_end + 1, bw_end + 1); seg("fwd head", bw_end + 1, len(ms))
# side streams: when do they finish relative to the main stream's last kernel before the solver
f.write("# main stream last kernel ends at %.1f us; per stream last end: %s\n" % ((ms[-1][2] - t0) / 1e3, ", ".join("%s %.1f" % (s, (max(r[2] for r in rs) - t0) / 1e3) for s, rs in by.items())))
# the main stream's kernels of the backward phase by name
agg = {}
for r in ms[fw_end + 1:bw_end + 1]:
    k = (r[0][:44], "%dx%dx%d" % (r[4], r[5], r[6])); a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += r[2] - r[1]
f.write("# backward phase of the main stream by kernel / grid\n")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    f.write("%-46s %18s %5d %10.1f us\n" % (k[0], k[1], n, d / 1e3))
agg = {}
for r in ms[:fw_end + 1] + ms[bw_end + 1:]:
    k = (r[0][:44], "%dx%dx%d" % (r[4], r[5], r[6])); a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += r[2] - r[1]
f.write("# forward phase of the main stream by kernel / grid\n")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    f.write("%-46s %18s %5d %10.1f us\n" % (k[0], k[1], n, d / 1e3))
f.close()
print(open(out).read())
