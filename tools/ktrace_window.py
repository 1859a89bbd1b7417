"""developer tool: start/end of the dispatches around the n-th launch of a kernel (overlap check)
usage: ktrace_window.py <kernel-substring> <n> -- <command>"""
import glob, os, shutil, sqlite3, subprocess, sys, tempfile
cut = sys.argv.index("--")
pat, nth = sys.argv[1], int(sys.argv[2])
cmd = [os.path.abspath(c) if os.path.exists(c) and c.endswith(".py") else c for c in sys.argv[cut + 1:]]
out = tempfile.mkdtemp(prefix="ktw_", dir="/tmp")
subprocess.call(["rocprofv3", "--kernel-trace", "-d", out, "-o", "k", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                stdout=open(os.path.join(out, "log"), "w"), stderr=subprocess.STDOUT)
db = sqlite3.connect(glob.glob(os.path.join(out, "**", "*.db"), recursive=True)[0])
cur = db.cursor()
sfx = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id=s.id order by d.start" % (sfx, sfx)))
idx = [i for i, r in enumerate(rows) if pat in r[0]]
i0 = idx[nth]
t0 = rows[i0][1]
for r in rows[max(0, i0 - 3):i0 + 25]:
    print("%-46s q%-3s start %9.3f ms  end %9.3f ms  (%.3f ms)" % (r[0][:46], r[3], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6))
shutil.rmtree(out, ignore_errors=True)
