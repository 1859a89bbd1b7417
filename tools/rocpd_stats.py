"""Per-kernel summary of a rocprofv3 trace database (rocpd sqlite, the default output of rocprofv3 7.x):
python tools/rocpd_stats.py <results.db> [top]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    dis = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    q = ("select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (dis, sym))
    rows = cur.execute(q).fetchall()
    tot = sum(r[2] for r in rows)
    for name, n, t, lo, hi in rows[:top]:
        print("  %-70s %6d x %10.3f ms total %9.4f ms avg  (min %.4f max %.4f)  %5.1f %%"
              % (name[:70], n, t / 1e6, t / n / 1e6, lo / 1e6, hi / 1e6, 100.0 * t / tot))


main()
