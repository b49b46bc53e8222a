"""Per-shape view of a rocprofv3 (rocpd sqlite) kernel trace: kernels grouped by (name, grid, workgroup) with average duration,
plus the launch sequence of the last `tail` dispatches with the idle gap in front of each (graph replay: what a step really costs)."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 400
c = sqlite3.connect(db)
cur = c.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
def pick(*names):
    for n in names:
        if n in cols: return n
    return None
gx, gy, gz = pick("grid_x", "grid_size_x"), pick("grid_y", "grid_size_y"), pick("grid_z", "grid_size_z")
wx = pick("workgroup_x", "workgroup_size_x")
st, en = pick("start"), pick("end")
with open(out, "w") as f:
    f.write("# columns of the kernels view: %s\n" % ",".join(cols))
    q = "select name, %s, %s, %s, %s, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels group by 1,2,3,4,5 order by 10 desc" % (gx, gy, gz, wx)
    rows = c.execute(q).fetchall()
    tot = sum(r[9] for r in rows)
    f.write("# %-46s %22s %5s %6s %10s %10s %10s %6s\n" % ("kernel", "grid(threads)", "wg", "calls", "avg_us", "min_us", "max_us", "pct"))
    for n, a, b, d, w, k, av, mn, mx, s in rows:
        if s < 0.0005 * tot: continue
        f.write("%-48s %22s %5d %6d %10.1f %10.1f %10.1f %5.2f%%\n" % (n[:48], "%dx%dx%d" % (a, b, d), w, k, av / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    seq = c.execute("select name, %s, %s, %s, %s, %s, duration from kernels order by %s" % (gx, gy, gz, st, en, st)).fetchall()
    seq = seq[-tail:]
    f.write("\n# last %d dispatches in start order: gap_us = start - max(end of all earlier dispatches)\n" % len(seq))
    last_end = None; busy = 0; gaps = 0
    for n, a, b, d, s, e, du in seq:
        gap = 0.0 if last_end is None else (s - last_end) / 1e3
        f.write("%-40s %18s %9.1f %8.1f\n" % (n[:40], "%dx%dx%d" % (a, b, d), du / 1e3, gap))
        if last_end is not None and gap > 0: gaps += gap
        busy += du / 1e3
        last_end = e if last_end is None else max(last_end, e)
    f.write("# sum of durations %.1f us, sum of positive gaps %.1f us, span %.1f us\n" % (busy, gaps, (last_end - seq[0][4]) / 1e3))
print(open(out).read()[:6000])
