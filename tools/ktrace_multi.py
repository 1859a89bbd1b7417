"""developer tool: merged kernel timeline of a multi-process run (one rocprofv3 database per rank): the dispatches of
all ranks around the n-th launch of a kernel, and per-kernel totals per rank.
usage: ktrace_multi.py <kernel-substring> <n> [rows] -- <command>"""
import glob, os, shutil, sqlite3, subprocess, sys, tempfile
cut = sys.argv.index("--")
pat, nth = sys.argv[1], int(sys.argv[2])
nrows = int(sys.argv[3]) if cut > 3 else 60
cmd = [os.path.abspath(c) if os.path.exists(c) and c.endswith(".py") else c for c in sys.argv[cut + 1:]]
out = tempfile.mkdtemp(prefix="ktm_", dir="/tmp")
subprocess.call(["rocprofv3", "--kernel-trace", "-d", out, "-o", "k_%pid%", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                stdout=open(os.path.join(out, "log"), "w"), stderr=subprocess.STDOUT)
print("\n".join(l for l in open(os.path.join(out, "log")).read().splitlines() if l.startswith("[bench] stages")))
rows = []
for pi, f in enumerate(sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True))):
    db = sqlite3.connect(f)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
    if not tabs:
        continue
    sfx = tabs[0].replace("rocpd_kernel_dispatch", "")
    got = list(cur.execute("select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id=s.id" % (sfx, sfx)))
    if not got:
        continue
    rows += [(r[1], r[2], pi, r[3], r[0]) for r in got]
    tot = {}
    for r in got:
        k = r[0].split("(")[0][:50]
        a = tot.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e6
    print("---- process %d (%s): %d dispatches" % (pi, os.path.basename(f), len(got)))
    for k, a in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-50s n=%6d total %9.3f ms  avg %8.4f ms" % (k, a[0], a[1], a[1] / a[0]))
rows.sort()
idx = [i for i, r in enumerate(rows) if pat in r[4]]
if idx:
    i0 = idx[min(nth, len(idx) - 1)]
    t0 = rows[i0][0]
    for r in rows[max(0, i0 - 4):i0 + nrows]:
        print("P%d q%-3s %-44s start %9.3f  end %9.3f  (%.3f ms)" % (r[2], r[3], r[4].split("(")[0][:44], (r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e6))
shutil.rmtree(out, ignore_errors=True)
