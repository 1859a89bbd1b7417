"""Per-kernel averages of a few SQ counters (rocprofv3 --pmc with --kernel-trace only; one pass per counter group).
usage: python tools/pmc_sq.py <kernel-name-substring> -- <command ...>"""
import csv, glob, os, shutil, subprocess, sys, tempfile, collections

GROUPS = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
          ["SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"],
          ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM_WR", "SQ_LDS_BANK_CONFLICT"],
          ["SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SMEM", "SQ_LDS_IDX_ACTIVE"]]


def one_pass(counters, cmd, pat):
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.call(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "-d", out, "-o", "p", "--output-format", "csv", "--"] + cmd,
                    cwd="/tmp", env=env, stdout=open(os.path.join(out, "log"), "w"), stderr=subprocess.STDOUT)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for q in pat.split(","):
                if q in r["Kernel_Name"]:
                    a = acc[(q, r["Counter_Name"])]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    shutil.rmtree(out, ignore_errors=True)
    return acc


def main():
    cut = sys.argv.index("--")
    pat, cmd = sys.argv[1], sys.argv[cut + 1:]
    cmd = [os.path.abspath(c) if c.endswith(".py") and os.path.exists(c) else c for c in cmd]
    groups = GROUPS if not os.environ.get("PMC_GROUPS") else [g.split(",") for g in os.environ["PMC_GROUPS"].split(";")]
    for g in groups:
        acc = one_pass(g, cmd, pat)
        for q in pat.split(","):
            for c in g:
                n, v = acc.get((q, c), [0, 0.0])
                print("%-16s %-28s launches %5d  avg %.4g" % (q, c, n, v / max(n, 1)), flush=True)


if __name__ == "__main__":
    main()
