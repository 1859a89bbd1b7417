"""Where do the small device copies of a bench step come from? (developer tool)  Counts hipMemcpy-class calls of the library
per Python call site during one step of a workload: python tools/prof_copies.py cfg5"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tigar_amd as t
from tigar_amd import device as dev, _lib

counts = collections.Counter()
L = _lib.lib()
names = [n for n in ("tg_vec_from_host", "tg_vec_to_host", "tg_csr_from_host", "tg_csr_download", "tg_vec_copy", "tg_vec_create",
                     "tg_csr_download_rows", "tg_vec_set_range", "tg_vec_get_range", "tg_eval_basis_1d") if hasattr(L, n)]
for n in names:
    f = getattr(L, n)
    def wrap(*a, _f=f, _n=n):
        st = traceback.extract_stack(limit=6)
        site = " <- ".join("%s:%d" % (os.path.basename(s.filename), s.lineno) for s in st[-5:-1])
        counts[(_n, site)] += 1
        return _f(*a)
    wrap.argtypes = getattr(f, "argtypes", None)
    wrap.restype = getattr(f, "restype", None)
    setattr(L, n, wrap)
sys.argv = ["bench.py", "--workload", sys.argv[1] if len(sys.argv) > 1 else "cfg5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--rtol", "1e-10"]
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
for (n, site), c in counts.most_common(25):
    print("%6d  %-22s %s" % (c, n, site), file=sys.stderr)
