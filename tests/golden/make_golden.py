#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory from the REFERENCE's own source,
executed through the stub import of ``_ref_stub_import.py`` (works only in the build
container, where /root/reference exists).  Run:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are data only (inputs + the reference's outputs):

  golden_knots.npz      uniformKnots                         tIGAr/BSplines.py:14-38
  golden_bspline1.npz   BSpline1 bookkeeping, getKnotSpan, getNodes, basisFuncs, greville
                                                             tIGAr/BSplines.py:164-351
  golden_tensor.npz     BSpline.getNodesAndEvals tables on the canonical Q_p node grid,
                        the resulting extraction matrix M (eps filter + sorted columns
                        applied to the reference rows as generateM does,
                        tIGAr/common.py:1554-1571), getSideDofs, Greville control points,
                        getPrealloc/getDegree/needsDG          tIGAr/BSplines.py:374-649,910-963
"""
import os, sys, json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_stub_import import import_reference  # noqa: E402

B = import_reference()


def fe_nodes_1d(spline1, p):
    """Canonical node grid (see oracle/tigar_oracle.py:fe_nodes_1d): vertices exactly on
    the reference's uniqueKnots, interior nodes x0*(1-t)+x1*t, t=j/p."""
    uk = spline1.uniqueKnots
    nel = spline1.nel
    x = np.empty(nel * p + 1)
    for e in range(nel):
        for j in range(p):
            t = float(j) / float(p)
            x[e * p + j] = uk[e] * (1.0 - t) + uk[e + 1] * t
    x[nel * p] = uk[nel]
    return x


def gen_knots():
    out = {}
    meta = []
    k = 0
    for p in (1, 2, 3, 4):
        for N in (1, 4, 10, 32):
            for periodic in (False, True):
                for drop in (0, 1):
                    if drop >= p:
                        continue
                    for (a, b) in ((0.0, 1.0), (-1.0, 1.0), (0.3, 2.7)):
                        kv = B.uniformKnots(p, a, b, N, periodic, drop)
                        out["k%d" % k] = np.array(kv, dtype=np.float64)
                        meta.append([p, a, b, N, int(periodic), drop])
                        k += 1
    out["meta"] = np.array(meta, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "golden_knots.npz"), **out)
    print("golden_knots:", k, "cases")


def knot_cases_1d():
    cases = []
    for p in (1, 2, 3, 4):
        for N in (1, 2, 4, 8):
            cases.append((p, B.uniformKnots(p, 0.0, 1.0, N)))
        cases.append((p, B.uniformKnots(p, -1.0, 1.0, 7)))
    # non-uniform open
    cases.append((2, [0, 0, 0, 0.1, 0.35, 0.4, 0.9, 1, 1, 1]))
    cases.append((3, [0, 0, 0, 0, 0.2, 0.25, 0.7, 1.5, 2, 2, 2, 2]))
    # repeated interior knots (reduced continuity)
    cases.append((2, B.uniformKnots(2, 0.0, 1.0, 4, False, 1)))
    cases.append((3, B.uniformKnots(3, 0.0, 1.0, 4, False, 1)))
    cases.append((3, [0, 0, 0, 0, 0.5, 0.5, 0.5, 1, 1, 1, 1]))       # C0 at 0.5
    cases.append((2, [0, 0, 0, 0.5, 0.5, 0.5, 1, 1, 1]))              # discontinuous
    # periodic
    cases.append((2, B.uniformKnots(2, 0.0, 1.0, 6, True)))
    cases.append((3, B.uniformKnots(3, 0.0, 2.0, 8, True)))
    return cases


def gen_bspline1():
    out = {}
    cases = knot_cases_1d()
    for ci, (p, kv) in enumerate(cases):
        s = B.BSpline1(p, kv)
        pre = "c%d_" % ci
        out[pre + "p"] = np.array(p)
        out[pre + "knots"] = np.array(kv, dtype=np.float64)
        out[pre + "uniqueKnots"] = np.array(s.uniqueKnots)
        out[pre + "multiplicities"] = np.array(s.multiplicities)
        out[pre + "nel"] = np.array(s.nel)
        out[pre + "ncp"] = np.array(s.ncp)
        out[pre + "ghostKnots"] = np.array(s.ghostKnots)
        out[pre + "disc"] = np.array(int(s.isDiscontinuous()))
        out[pre + "greville"] = np.array([s.greville(i) for i in range(s.ncp)])
        # evaluation points: knots, +-1ulp, span interiors, FE nodes for degree p
        us = []
        for u in s.uniqueKnots:
            us += [u, np.nextafter(u, -np.inf), np.nextafter(u, np.inf)]
        for e in range(s.nel):
            a, b = s.uniqueKnots[e], s.uniqueKnots[e + 1]
            us += [a + (b - a) * t for t in (0.5, 0.123, 0.987, 1.0 / 3.0)]
        us += list(fe_nodes_1d(s, p))
        us = np.array([u for u in us if s.knots[0] <= u <= s.knots[-1]])
        spans = np.array([s.getKnotSpan(u) for u in us], dtype=np.int64)
        nodes = np.array([s.getNodes(u) for u in us], dtype=np.int64)
        ders = np.array([s.basisFuncs(int(sp_), u) for sp_, u in zip(spans, us)])
        out[pre + "u"] = us
        out[pre + "span"] = spans
        out[pre + "nodes"] = nodes
        out[pre + "ders"] = ders
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "golden_bspline1.npz"), **out)
    print("golden_bspline1:", len(cases), "cases")


def tensor_cases():
    U = B.uniformKnots
    cases = []
    # (name, degrees, kvecs)
    for p in (2, 3, 4):
        for nel in (1, 2, 4):
            cases.append(("1d_p%d_n%d" % (p, nel), [p], [U(p, 0., 1., nel)]))
    for p in (1, 2, 3, 4):
        for nel in (1, 2, 4, 8):
            if p == 4 and nel == 8:
                continue
            cases.append(("2d_p%d_n%d" % (p, nel), [p, p], [U(p, 0., 1., nel)] * 2))
    for p in (2, 3):
        for nel in (1, 2, 4):
            cases.append(("3d_p%d_n%d" % (p, nel), [p, p, p], [U(p, 0., 1., nel)] * 3))
    cases.append(("3d_p4_n2", [4, 4, 4], [U(4, 0., 1., 2)] * 3))
    # anisotropic counts / domains, equal degree
    cases.append(("2d_p2_5x3", [2, 2], [U(2, -1., 1., 5), U(2, 0., 3., 3)]))
    cases.append(("3d_p2_3x4x2", [2, 2, 2], [U(2, 0., 1., 3), U(2, -1., 1., 4), U(2, 0., 2., 2)]))
    # mixed degrees (FE degree = max, tIGAr/BSplines.py:580-588)
    cases.append(("2d_p23_n3", [2, 3], [U(2, 0., 1., 3), U(3, 0., 1., 3)]))
    # non-uniform + reduced continuity
    cases.append(("2d_nonuni", [2, 3], [[0, 0, 0, 0.1, 0.35, 0.4, 0.9, 1, 1, 1],
                                        [0, 0, 0, 0, 0.5, 0.5, 0.5, 1, 1, 1, 1]]))
    cases.append(("2d_p3_drop1", [3, 3], [U(3, 0., 1., 3, False, 1)] * 2))
    # periodic in one direction (index wrap, tIGAr/BSplines.py:318)
    cases.append(("2d_periodic", [2, 2], [U(2, 0., 1., 6, True), U(2, 0., 1., 3)]))
    # biharmonic-demo-like: (-1,1)^2 p=4
    cases.append(("2d_p4_bih", [4, 4], [U(4, -1., 1., 3)] * 2))
    return cases


def gen_tensor():
    out = {}
    names = []
    eps = 1e-15
    for name, degs, kvecs in tensor_cases():
        s = B.BSpline(degs, kvecs)
        cm = B.ExplicitBSplineControlMesh(degs, kvecs)
        d = len(degs)
        deg = s.getDegree()
        axes = [fe_nodes_1d(s.splines[k], deg) for k in range(d)]
        n = [len(a) for a in axes]
        nrows = int(np.prod(n))
        ne0 = len(s.getNodesAndEvals([axes[k][0] for k in range(d)]))
        cols = np.empty((nrows, ne0), dtype=np.int64)
        vals = np.empty((nrows, ne0))
        rowptr = [0]
        mc = []
        mv = []
        for r in range(nrows):
            idx = []
            rr = r
            for k in range(d):
                idx.append(rr % n[k])
                rr //= n[k]
            xi = [axes[k][idx[k]] for k in range(d)]
            ne = s.getNodesAndEvals(xi)
            cols[r] = [int(e[0]) for e in ne]
            vals[r] = [float(e[1]) for e in ne]
            # generateM loop body, tIGAr/common.py:1566-1571 (INSERT => last wins), then
            # PETSc assembly sorts the row by column
            row = {}
            for c, v in ne:
                if abs(v) > eps:
                    row[int(c)] = float(v)
            for c in sorted(row):
                mc.append(c)
                mv.append(row[c])
            rowptr.append(len(mc))
        pre = name + "/"
        out[pre + "degrees"] = np.array(degs)
        for k in range(d):
            out[pre + "kvec%d" % k] = np.array(kvecs[k], dtype=np.float64)
            out[pre + "axis%d" % k] = axes[k]
        out[pre + "ne_cols"] = cols.astype(np.int32)
        out[pre + "ne_vals"] = vals
        out[pre + "M_rowptr"] = np.array(rowptr, dtype=np.int64)
        out[pre + "M_col"] = np.array(mc, dtype=np.int32)
        out[pre + "M_val"] = np.array(mv, dtype=np.float64)
        out[pre + "ncp"] = np.array(s.getNcp())
        out[pre + "prealloc"] = np.array(s.getPrealloc())
        out[pre + "degree"] = np.array(deg)
        out[pre + "needsDG"] = np.array(int(s.needsDG()))
        for direction in range(d):
            for side in (0, 1):
                for nl in (1, 2):
                    out[pre + "side_%d_%d_%d" % (direction, side, nl)] = \
                        np.array(s.getSideDofs(direction, side, nl), dtype=np.int64)
        nsd = cm.getNsd()
        P = np.array([[cm.getHomogeneousCoordinate(I, j) for j in range(nsd + 1)]
                      for I in range(s.getNcp())])
        out[pre + "P"] = P
        names.append(name)
        print("  ", name, "rows", nrows, "nnz", len(mc))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "golden_tensor.npz"), **out)
    print("golden_tensor:", len(names), "cases")


if __name__ == "__main__":
    gen_knots()
    gen_bspline1()
    gen_tensor()
