"""developer tool: where the GPU idles inside a run (host gaps between dispatches).

    python tools/gpu_gaps.py [--from KERNEL_SUBSTRING --nth N] [--top 25] -- <command>

Runs the command under `rocprofv3 --kernel-trace`, orders the dispatches by start time and reports, for the window that
starts at the N-th launch (default: the last but one) of a kernel whose name contains KERNEL_SUBSTRING and ends at its next
launch -- one step of bench.py when the kernel runs once per step --, the busy time (union of the dispatch intervals), the
idle time, and the largest gaps with the kernels on either side."""
import argparse, glob, os, shutil, sqlite3, subprocess, sys, tempfile


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="frm", default=None)
    ap.add_argument("--nth", type=int, default=-2)
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args(argv[:cut])
    cmd = [os.path.abspath(c) if os.path.exists(c) and c.endswith(".py") else c for c in argv[cut + 1:]]
    out = tempfile.mkdtemp(prefix="gaps_", dir="/tmp")
    subprocess.call(["rocprofv3", "--kernel-trace", "-d", out, "-o", "k", "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                    stdout=open(os.path.join(out, "log"), "w"), stderr=subprocess.STDOUT)
    db = sqlite3.connect(glob.glob(os.path.join(out, "**", "*.db"), recursive=True)[0])
    cur = db.cursor()
    sfx = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s "
                            "on d.kernel_id=s.id order by d.start" % (sfx, sfx)))
    lo, hi = 0, len(rows)
    if args.frm:
        idx = [i for i, r in enumerate(rows) if args.frm in r[0]]
        lo = idx[args.nth]
        nxt = [i for i in idx if i > lo]
        hi = nxt[0] if nxt else len(rows)
    win = rows[lo:hi]
    t0, t1 = win[0][1], max(r[2] for r in win)
    busy, gaps, end = 0.0, [], win[0][1]
    prev = win[0][0]
    for name, s, e in win:
        if s > end:
            gaps.append((s - end, prev, name, end - t0))
            busy += e - s
        else:
            busy += max(0, e - max(s, end))
        if e > end:
            end, prev = e, name
    print("window: %d dispatches, %.3f ms; busy %.3f ms, idle %.3f ms (%.1f %%)"
          % (len(win), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / max(1, t1 - t0)))
    small = sum(g[0] for g in gaps if g[0] < 20e3)
    print("gaps: %d, of which %d shorter than 20 us (%.3f ms together)" % (len(gaps), sum(1 for g in gaps if g[0] < 20e3), small / 1e6))
    for g in sorted(gaps, reverse=True)[:args.top]:
        print("  %8.3f ms at %9.3f ms   after %-44s before %-44s" % (g[0] / 1e6, g[3] / 1e6, g[1][:44], g[2][:44]))
    shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main()
