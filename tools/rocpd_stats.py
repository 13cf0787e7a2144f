"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / min / max duration [us].
Usage: python tools/rocpd_stats.py <results.db> [grid-filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
q = ("select name, grid_x, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) "
     "from kernels group by name, grid_x order by sum(duration) desc")
print("%-58s %9s %6s %9s %9s %9s %5s %6s" % ("kernel", "grid_x", "calls", "avg_us", "min_us", "max_us", "vgpr", "lds"))
for name, grid, n, avg, mn, mx, vg, lds in cur.execute(q):
    short = name.split("(")[0][-58:]
    print("%-58s %9d %6d %9.2f %9.2f %9.2f %5d %6d" % (short, grid, n, avg / 1e3, mn / 1e3, mx / 1e3, vg, lds))
