#!/usr/bin/env python3
"""
Fixture + golden vectors for the Rhino T-spline reader (in-container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rhino.py

1. writes ``tspline_bicubic_patch.iga``: a synthetic file in the Rhino element-extraction format, built here from
   a bicubic B-spline patch (non-uniform knots, rational weights) by Bezier extraction -- every Bezier element lists
   its 16 functions and their Bernstein coefficients; the functions of some elements are listed in scrambled order,
   and exact-zero rows are appended to one element, as real exports contain;
2. reads it with the REFERENCE's ``RhinoTSplineScalarBasis`` / ``RhinoTSplineControlMesh`` (tIGAr/RhinoTSplines.py,
   stub import) and stores, for every FE node of the disconnected degree-3 element mesh (node coordinates in the
   file ``golden_rhino.npz``), the reference's ``getNodesAndEvals`` output (function indices and values, in the
   reference's order), plus ncp / nelBez / maxNshl / degree and the homogenised control net.
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_stub_import as R  # noqa: E402

B = R.import_reference()
import tIGAr.RhinoTSplines as RT  # noqa: E402
from oracle import tigar_oracle as O  # noqa: E402

FNAME = os.path.join(HERE, "tspline_bicubic_patch.iga")


def bezier_extraction_1d(knots, p=3):
    """per element: C1[e][local function][Bernstein index] of the 1-D B-spline with ``knots``"""
    s = O.BSpline1(p, knots)
    uk = s.uniqueKnots
    t = np.array([0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0])
    bern = np.array([[(1 - x) ** 3, 3 * x * (1 - x) ** 2, 3 * x ** 2 * (1 - x), x ** 3] for x in t])   # [point][b]
    out = []
    for e in range(len(uk) - 1):
        xs = uk[e] + (uk[e + 1] - uk[e]) * t
        xs[0] += 1e-13 * (uk[e + 1] - uk[e])          # stay inside the element (left-biased span at knots)
        vals = np.zeros((4, 4))                        # [point][local function]
        first = None
        for q, x in enumerate(xs):
            sp_ = s.getKnotSpan(x)
            nodes = s.getNodes(x)
            first = nodes[0] if first is None else first
            assert nodes[0] == first
            vals[q] = s.basisFuncs(sp_, x)
        C = np.linalg.solve(bern, vals).T              # [function][b]: vals = bern @ C^T
        C[np.abs(C) < 1e-14] = 0.0
        out.append((first, C))
    return out, s.getNcp()


def write_file():
    kx = [0, 0, 0, 0, 0.3, 0.55, 1, 1, 1, 1]
    ky = [0, 0, 0, 0, 0.6, 1, 1, 1, 1]
    Cx, nx = bezier_extraction_1d(kx)
    Cy, ny = bezier_extraction_1d(ky)
    rng = np.random.default_rng(12)
    lines = ["type surface", "nodeN %d" % (nx * ny), "elemN %d" % (len(Cx) * len(Cy))]
    for j in range(ny):
        for i in range(nx):
            w = 0.8 + 0.4 * rng.random()
            lines.append("node %.17g %.17g %.17g %.17g" % (i / (nx - 1.0) + 0.05 * rng.standard_normal(), j / (ny - 1.0),
                                                          0.1 * np.sin(i + j), w))
    ecount = 0
    for ey, (fy, C1y) in enumerate(Cy):
        for ex, (fx, C1x) in enumerate(Cx):
            nodes, rows = [], []
            for jl in range(4):
                for il in range(4):
                    nodes.append((fx + il) + nx * (fy + jl))
                    rows.append([C1y[jl][j] * C1x[il][i] for j in range(4) for i in range(4)])
            order = list(range(16))
            if ecount % 2 == 1:
                order = list(rng.permutation(16))          # scrambled listing order
            nodes = [nodes[o] for o in order]
            rows = [rows[o] for o in order]
            if ecount == 2:                                 # a function with an all-zero row on this element
                extra = [n for n in range(nx * ny) if n not in nodes][0]
                nodes.append(extra)
                rows.append([0.0] * 16)
            lines.append("belem %d 3 3" % len(nodes))
            lines.append(" ".join(str(n) for n in nodes))
            for r in rows:
                lines.append(" ".join("%.17g" % v for v in r))
            ecount += 1
    with open(FNAME, "w") as f:
        f.write("\n".join(lines) + "\n")
    return nx * ny, ecount


def main():
    ncp, nel = write_file()
    basis = RT.RhinoTSplineScalarBasis(FNAME)
    cm = RT.RhinoTSplineControlMesh(FNAME)
    assert basis.getNcp() == ncp and basis.nelBez == nel
    # FE nodes: degree-3 Lagrange nodes of the disconnected elements [3e, 3e+2] x [-1, 1], x fastest, element-major
    t = np.arange(4) / 3.0
    X = []
    for e in range(nel):
        x0, x1 = 3.0 * e, 3.0 * e + 2.0
        xs = x0 * (1.0 - t) + x1 * t
        xs[0], xs[-1] = x0, x1
        ys = -1.0 * (1.0 - t) + 1.0 * t
        ys[0], ys[-1] = -1.0, 1.0
        for y in ys:
            for x in xs:
                X.append([x, y])
    X = np.array(X)
    cnt, nodes, vals = [], [], []
    for xi in X:
        ne = basis.getNodesAndEvals(xi)
        cnt.append(len(ne))
        nodes += [int(a[0]) for a in ne]
        vals += [float(a[1]) for a in ne]
    np.savez_compressed(os.path.join(HERE, "golden_rhino.npz"), X=X, cnt=np.array(cnt, dtype=np.int64),
                        nodes=np.array(nodes, dtype=np.int64), vals=np.array(vals), ncp=np.int64(basis.getNcp()),
                        nel=np.int64(basis.nelBez), maxNshl=np.int64(basis.getPrealloc()), degree=np.int64(basis.getDegree()),
                        bnet=np.asarray(cm.bnet), nsd=np.int64(cm.getNsd()))
    print("wrote", FNAME, "and golden_rhino.npz:", ncp, "functions,", nel, "elements,", len(X), "FE nodes")


if __name__ == "__main__":
    main()
