"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (from %s)\n" % db.split("/")[-1])
    f.write("# %-100s %8s %14s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for n, k, s, a, mn, mx in rows:
        f.write("%-102s %8d %14d %12.0f %12d %12d %6.2f%%\n" % (n[:100], k, s, a, mn, mx, 100.0 * s / tot))
print(open(out).read())
