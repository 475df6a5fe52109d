"""Per-kernel totals from a rocprofv3 rocpd (sqlite) database: python tools/rocpd_stats.py <results.db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t and "rocpd_info" in t][0]
rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
                  "on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print("kernel,calls,total_ms,percent,avg_us,min_us,max_us")
for r in rows[:top]:
    print("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f" % (r[0].replace(",", ";")[:120], r[1], r[2] / 1e6, 100.0 * r[2] / tot, r[2] / r[1] / 1e3,
                                             r[3] / 1e3, r[4] / 1e3))
print("TOTAL,,%.3f" % (tot / 1e6))
