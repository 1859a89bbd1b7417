#!/usr/bin/env python3
"""
Golden fixture for multi-patch B-splines from the REFERENCE's ``MultiBSpline`` (tIGAr/BSplines.py:651-908; stub import,
in-container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_multipatch.py

golden_multipatch.npz holds, per seeded random configuration (2-4 bivariate patches of one degree pair, random element counts
and knot ranges): the inputs (degrees, the knot vectors as given), and the reference's outputs -- normalised knot vectors,
doffsets, ncp, nel, getPatchSideDofs for every patch / direction / side / one and two layers, and getNodesAndEvals (columns in
the reference's order, values) at sample points of every element of every patch (corners, edge and interior points, in the
global coordinates of the multi-patch mesh: patch k occupies [2k, 2k+1] x [0, 1]).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_stub_import import import_reference  # noqa: E402

B = import_reference()


def main():
    rng = np.random.default_rng(77)
    out, meta = {}, []
    for ci in range(10):
        npatch = int(rng.integers(2, 5))
        degs = [int(rng.integers(1, 4)), int(rng.integers(1, 4))]
        kvs, patches = [], []
        for k in range(npatch):
            kv = []
            for d in range(2):
                a = float(rng.choice([0.0, -1.0, 0.5]))
                b = a + float(rng.choice([1.0, 2.0, 3.5]))
                kv.append(B.uniformKnots(degs[d], a, b, int(rng.integers(1, 5)), False, int(rng.integers(0, degs[d])) if rng.random() < 0.3 else 0))
            kvs.append(kv)
            patches.append(B.BSpline(degs, [list(v) for v in kv]))
        mb = B.MultiBSpline(patches)
        name = "mp%02d" % ci
        meta.append({"name": name, "npatch": npatch, "degrees": degs})
        out[name + "_doffsets"] = np.array(mb.doffsets, dtype=np.int64)
        out[name + "_ncp"] = np.int64(mb.getNcp())
        out[name + "_nel"] = np.int64(mb.nel)
        pts, cols, vals, ptr = [], [], [], [0]
        for k in range(npatch):
            for d in range(2):
                out["%s_p%d_kv%d_in" % (name, k, d)] = np.array(kvs[k][d], dtype=np.float64)
                out["%s_p%d_kv%d_norm" % (name, k, d)] = np.array(patches[k].splines[d].knots, dtype=np.float64)
                for side in (0, 1):
                    for nl in (1, 2):
                        out["%s_p%d_side_%d_%d_%d" % (name, k, d, side, nl)] = \
                            np.array(mb.getPatchSideDofs(k, d, side, nl), dtype=np.int64)
            us, vs = patches[k].splines[0], patches[k].splines[1]
            for i in range(us.nel):
                for j in range(vs.nel):
                    for tu in (0.0, 0.3, 1.0):
                        for tv in (0.0, 0.55, 1.0):
                            x = 2.0 * k + us.uniqueKnots[i] * (1.0 - tu) + us.uniqueKnots[i + 1] * tu
                            y = vs.uniqueKnots[j] * (1.0 - tv) + vs.uniqueKnots[j + 1] * tv
                            ne = mb.getNodesAndEvals(np.array([x, y]))
                            pts.append([x, y])
                            cols += [int(e[0]) for e in ne]
                            vals += [float(e[1]) for e in ne]
                            ptr.append(len(cols))
        out[name + "_pts"] = np.array(pts)
        out[name + "_ptr"] = np.array(ptr, dtype=np.int64)
        out[name + "_cols"] = np.array(cols, dtype=np.int64)
        out[name + "_vals"] = np.array(vals)
        print("  ", name, "patches", npatch, "degrees", degs, "points", len(pts))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "golden_multipatch.npz"), **out)


if __name__ == "__main__":
    main()
