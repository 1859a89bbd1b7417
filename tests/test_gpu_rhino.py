"""-m gpu: extraction of a Rhino T-spline (tg_extract_csr_bezier, csrc/tg_bezier.hip) through the generator API
against the reference's own per-node output (golden_rhino.npz: bit-exact after the reference's abs(v) > eps filter and
PETSc's column ordering), against the B-spline the synthetic file was derived from, and through extractMatrix."""
import os
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FNAME = os.path.join(GOLDEN, "tspline_bicubic_patch.iga")


def _golden_M(g, eps=1e-15):
    rows, cols, vals = [], [], []
    off = 0
    for r in range(len(g["cnt"])):
        n = int(g["cnt"][r])
        for c, v in zip(g["nodes"][off:off + n], g["vals"][off:off + n]):
            if abs(v) > eps:                     # generateM's filter (tIGAr/common.py:1569)
                rows.append(r), cols.append(int(c)), vals.append(float(v))
        off += n
    M = sp.csr_matrix((vals, (rows, cols)), shape=(len(g["cnt"]), int(g["ncp"])))
    M.sort_indices()
    return M


def test_rhino_tspline_extraction_matches_reference_output():
    import tigar_amd as t
    from tigar_amd.RhinoTSplines import RhinoTSplineControlMesh
    g = np.load(os.path.join(GOLDEN, "golden_rhino.npz"))
    cm = RhinoTSplineControlMesh(FNAME)
    gen = t.EqualOrderSpline(1, cm)
    M = gen.M.to_scipy()
    Mg = _golden_M(g)
    assert M.shape == Mg.shape == (96, 30)
    assert np.array_equal(M.indptr, Mg.indptr) and np.array_equal(M.indices, Mg.indices)      # pattern bit-exact
    assert np.array_equal(M.data, Mg.data)                                                     # values bit-exact
    # the generic plug-in path (host row loop -> tg_csr_from_triplets) gives the same matrix
    basis = cm.getScalarSpline()
    X = gen.V.grids[0].coordinates()
    rows, cols, vals = [], [], []
    for I in range(X.shape[0]):
        for c, v in basis.getNodesAndEvals(X[I]):
            rows.append(I), cols.append(c), vals.append(v)
    from tigar_amd import device as dev
    Mt = dev.csr_from_triplets(X.shape[0], 30, rows, cols, vals, 1e-15).to_scipy()
    assert np.array_equal(Mt.indices, M.indices) and np.array_equal(Mt.data, M.data)
    # the file was derived from a bicubic B-spline: rows are its basis values at the mapped element nodes
    kx = [0, 0, 0, 0, 0.3, 0.55, 1, 1, 1, 1]
    ky = [0, 0, 0, 0, 0.6, 1, 1, 1, 1]
    s = O.BSpline([3, 3], [kx, ky])
    ukx, uky = s.splines[0].uniqueKnots, s.splines[1].uniqueKnots
    nex = len(ukx) - 1
    dense = M.toarray()
    for r in range(96):
        e, n = divmod(r, 16)
        ex, ey = e % nex, e // nex
        u = (X[r, 0] - 3.0 * e) / 2.0
        v = (X[r, 1] + 1.0) / 2.0
        xi = np.array([ukx[ex] + (ukx[ex + 1] - ukx[ex]) * min(max(u, 1e-12), 1 - 1e-12),
                       uky[ey] + (uky[ey + 1] - uky[ey]) * min(max(v, 1e-12), 1 - 1e-12)])
        row = np.zeros(30)
        for c, val in s.getNodesAndEvals(xi):
            row[int(c)] += val
        assert np.max(np.abs(dense[r] - row)) < 1e-10
    # partition of unity and the control functions (weights enter through M_control * bnet)
    assert np.max(np.abs(dense.sum(axis=1) - 1.0)) < 1e-13
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    assert len(cp) == 4
    for c in range(4):
        assert np.max(np.abs(cp[c] - Mg @ g["bnet"][:, c])) < 1e-14
    # the path on top of it: M^T A M and M^T b with an arbitrary FE matrix (general PtAP kernel)
    rng = np.random.default_rng(3)
    A = sp.random(96, 96, density=0.2, random_state=5, format="csr") + sp.identity(96) * 3.0
    gen.addZeroDofs(0, [0, 1, 2, 29])
    spline = t.ExtractedSpline(gen, 6)
    K = spline.extractMatrix(A.tocsr()).to_scipy()
    Ko = O.extract_matrix(Mg, A.tocsr(), [0, 1, 2, 29])
    assert np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    b = rng.standard_normal(96)
    y = spline.extractVector(b).get_local()
    assert np.max(np.abs(y - O.extract_vector(Mg, b, [0, 1, 2, 29]))) <= 1e-13 * np.max(np.abs(b))
