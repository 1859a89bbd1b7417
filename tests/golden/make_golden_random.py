#!/usr/bin/env python3
"""
Golden fixtures for RANDOM tensor patches, from the REFERENCE's own source (stub import of ``_ref_stub_import.py``; works
only in the build container, where /root/reference exists).  The cases are drawn by the generator of the random parity runs
(``tests/fuzz/fuzz_parity.py: draw_case`` -- dimension, degrees per direction, element counts, periodic directions, repeated
knots by continuityDrop, non-uniform knots with random multiplicities), so that the oracle those runs compare with is pinned
to the reference on the same kind of input.  Run:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_random.py

  golden_random.npz   per case: degrees, knot vectors (uniform ones from the reference's uniformKnots), the extraction matrix
                      M on the canonical Q_p node grid (rows of BSpline.getNodesAndEvals with generateM's eps filter and
                      sorted columns, tIGAr/common.py:1554-1571), getSideDofs for one and two layers, getNcp / getDegree
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
from _ref_stub_import import import_reference  # noqa: E402
from make_golden import fe_nodes_1d  # noqa: E402
import fuzz_parity as fz  # noqa: E402

B = import_reference()


def main(ncases=56, seed=2024, max_rows=700):
    rng = np.random.default_rng(seed)
    out, names, metas = {}, [], []
    eps = 1e-15
    i = 0
    while len(names) < ncases:
        case = fz.draw_case(rng, max_rows)
        kvecs = fz.knot_vectors(case, B.uniformKnots)
        degs = case["ps"]
        try:
            s = B.BSpline(degs, kvecs)
        except Exception as e:          # (a draw the reference itself refuses)
            print("   skipped:", case, e)
            continue
        d = len(degs)
        deg = s.getDegree()
        axes = [fe_nodes_1d(s.splines[k], deg) for k in range(d)]
        n = [len(a) for a in axes]
        nrows = int(np.prod(n))
        rowptr, mc, mv = [0], [], []
        for r in range(nrows):
            idx, rr = [], r
            for k in range(d):
                idx.append(rr % n[k])
                rr //= n[k]
            ne = s.getNodesAndEvals([axes[k][idx[k]] for k in range(d)])
            row = {}
            for c, v in ne:
                if abs(v) > eps:
                    row[int(c)] = float(v)
            for c in sorted(row):
                mc.append(c)
                mv.append(row[c])
            rowptr.append(len(mc))
        name = "r%02d" % i
        i += 1
        pre = name + "/"
        out[pre + "degrees"] = np.array(degs)
        for k in range(d):
            out[pre + "kvec%d" % k] = np.array(kvecs[k], dtype=np.float64)
        out[pre + "M_rowptr"] = np.array(rowptr, dtype=np.int64)
        out[pre + "M_col"] = np.array(mc, dtype=np.int32)
        out[pre + "M_val"] = np.array(mv, dtype=np.float64)
        out[pre + "ncp"] = np.array(s.getNcp())
        out[pre + "degree"] = np.array(deg)
        for direction in range(d):
            for side in (0, 1):
                for nl in (1, 2):
                    out[pre + "side_%d_%d_%d" % (direction, side, nl)] = np.array(s.getSideDofs(direction, side, nl), dtype=np.int64)
        # evaluations at points that are NOT mesh nodes: random interior points, and per direction a unique knot with its two
        # floating-point neighbours (the span search's tie-breaking, tIGAr/BSplines.py:285-308)
        prng = np.random.default_rng(case["knot_seed"] + 1)
        lo = [float(s.splines[k].uniqueKnots[0]) for k in range(d)]
        hi = [float(s.splines[k].uniqueKnots[-1]) for k in range(d)]
        pts = [[lo[k] + (hi[k] - lo[k]) * float(prng.random()) for k in range(d)] for _ in range(16)]
        for k in range(d):
            uk = s.splines[k].uniqueKnots
            kn = float(uk[int(prng.integers(0, len(uk)))])
            for x in (kn, float(np.nextafter(kn, -np.inf)), float(np.nextafter(kn, np.inf))):
                if lo[k] <= x <= hi[k]:
                    q = [lo[j] + (hi[j] - lo[j]) * float(prng.random()) for j in range(d)]
                    q[k] = x
                    pts.append(q)
        ec, ev, ep = [], [], [0]
        for q in pts:
            ne = s.getNodesAndEvals(q)
            ec += [int(e[0]) for e in ne]
            ev += [float(e[1]) for e in ne]
            ep.append(len(ec))
        out[pre + "ev_pts"] = np.array(pts, dtype=np.float64).reshape(len(pts), d)
        out[pre + "ev_ptr"] = np.array(ep, dtype=np.int64)
        out[pre + "ev_cols"] = np.array(ec, dtype=np.int64)
        out[pre + "ev_vals"] = np.array(ev, dtype=np.float64)
        names.append(name)
        metas.append(json.dumps({k: case[k] for k in ("d", "ps", "kinds", "nels", "drops", "knot_seed")}))
        print("  ", name, metas[-1], "rows", nrows, "nnz", len(mc))
    out["names"] = np.array(names)
    out["meta"] = np.array(metas)
    np.savez_compressed(os.path.join(HERE, "golden_random.npz"), **out)
    print("golden_random:", len(names), "cases")


if __name__ == "__main__":
    main()
