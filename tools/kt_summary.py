"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): python tools/kt_summary.py <results.db>"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name "
                 "order by sum(end-start) desc").fetchall()
tot = sum(r[4] for r in rows)
print("%-100s %6s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "%"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-100s %6d %10.1f %10.1f %6.1f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, 100.0 * r[4] / tot))
