#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: prof_summary.py <results.db> [> profiles/xxx_kernel_stats.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
suffix = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""" % (suffix, suffix)
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print("%-78s %6s %10s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
for r in rows:
    print("%-78s %6d %10.3f %10.1f %10.1f %10.1f %6.1f" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
print("total kernel time: %.3f ms" % tot)
