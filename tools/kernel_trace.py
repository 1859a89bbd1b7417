"""Per-dispatch kernel durations of one command under rocprofv3 (kernel trace only).

    python tools/kernel_trace.py [--min-ms 1.0] [--sum] -- python bench.py ...

Prints every dispatch longer than --min-ms in launch order (name, ms, grid, LDS bytes), or with
--sum the per-kernel totals."""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-ms", type=float, default=1.0)
    ap.add_argument("--sum", action="store_true")
    args = ap.parse_args(argv[:cut])
    cmd = argv[cut + 1:]
    out = tempfile.mkdtemp(prefix="ktrace_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [os.path.abspath(c) if os.path.exists(c) and not os.path.isabs(c) and c.endswith(".py") else c for c in cmd]
    log = open(os.path.join(out, "cmd.log"), "w")
    rc = subprocess.call(["rocprofv3", "--kernel-trace", "-d", out, "-o", "k", "--"] + cmd, cwd="/tmp", env=env,
                         stdout=log, stderr=subprocess.STDOUT)
    log.close()
    tail = open(os.path.join(out, "cmd.log")).read().splitlines()[-6:]
    print("\n".join(t for t in tail if t.startswith("[bench]") or t.startswith("{")))
    dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
    if not dbs:
        print("no trace database (rc=%d)" % rc)
        return
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
    sfx = tabs[0].replace("rocpd_kernel_dispatch", "")
    q = ("select s.kernel_name, d.start, d.end-d.start, d.grid_size_x, d.group_segment_size "
         "from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id=s.id order by d.start" % (sfx, sfx))
    rows = list(cur.execute(q))
    if args.sum:
        tot = {}
        for name, st, dur, gx, lds in rows:
            t = tot.setdefault(name, [0, 0.0])
            t[0] += 1
            t[1] += dur / 1e6
        for name, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
            print("  %-60s %6d x  %10.3f ms total  %9.4f ms avg" % (name[:60], n, ms, ms / n))
    else:
        for name, st, dur, gx, lds in rows:
            if dur >= args.min_ms * 1e6:
                print("  %-52s %10.3f ms  grid %-11d lds %d" % (name[:52], dur / 1e6, gx, lds))
    shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main()
