"""Pins the oracle (oracle/tigar_oracle.py) against the golden vectors generated from the
reference's own source (tests/golden/make_golden.py).  CPU only."""
import os
import numpy as np
import pytest
from oracle import tigar_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_uniform_knots_bit_exact():
    g = _load("golden_knots.npz")
    meta = g["meta"]
    for k in range(meta.shape[0]):
        p, a, b, N, per, drop = meta[k]
        kv = O.uniform_knots(int(p), float(a), float(b), int(N), bool(per), int(drop))
        ref = g["k%d" % k]
        assert len(kv) == len(ref)
        assert np.array_equal(np.array(kv), ref)      # bit-exact


def test_bspline1_bookkeeping_and_evals_bit_exact():
    g = _load("golden_bspline1.npz")
    for ci in range(int(g["ncases"])):
        pre = "c%d_" % ci
        s = O.BSpline1(int(g[pre + "p"]), g[pre + "knots"])
        assert s.nel == int(g[pre + "nel"])
        assert s.ncp == int(g[pre + "ncp"])
        assert np.array_equal(s.uniqueKnots, g[pre + "uniqueKnots"])
        assert np.array_equal(s.multiplicities, g[pre + "multiplicities"])
        assert np.array_equal(s.ghostKnots, g[pre + "ghostKnots"])
        assert int(s.isDiscontinuous()) == int(g[pre + "disc"])
        assert np.array_equal(np.array([s.greville(i) for i in range(s.ncp)]),
                              g[pre + "greville"])
        us = g[pre + "u"]
        for q, u in enumerate(us):
            sp_ = s.getKnotSpan(u)
            assert sp_ == int(g[pre + "span"][q])
            assert s.getNodes(u) == list(g[pre + "nodes"][q])
            assert np.array_equal(s.basisFuncs(sp_, u), g[pre + "ders"][q])  # bit-exact


def _case_bspline(g, name):
    pre = name + "/"
    degs = [int(x) for x in g[pre + "degrees"]]
    kvecs = [g[pre + "kvec%d" % k] for k in range(len(degs))]
    return O.BSpline(degs, kvecs), pre


def test_tensor_nodes_and_evals_and_M_bit_exact():
    g = _load("golden_tensor.npz")
    for name in g["names"]:
        name = str(name)
        s, pre = _case_bspline(g, name)
        X, axes = O.fe_node_grid(s)
        for k in range(s.nvar):
            assert np.array_equal(axes[k], g[pre + "axis%d" % k])
        ne_cols = g[pre + "ne_cols"]
        ne_vals = g[pre + "ne_vals"]
        assert X.shape[0] == ne_cols.shape[0]
        step = max(1, X.shape[0] // 400)       # scalar path on a sample of rows
        for r in range(0, X.shape[0], step):
            ne = s.getNodesAndEvals(X[r])
            assert [int(e[0]) for e in ne] == list(ne_cols[r])
            assert np.array_equal(np.array([e[1] for e in ne]), ne_vals[r])
        # full M through the vectorised twin: pattern and values bit-exact
        M = O.generate_M_tensor(s)
        assert np.array_equal(M.indptr, g[pre + "M_rowptr"])
        assert np.array_equal(M.indices, g[pre + "M_col"])
        assert np.array_equal(M.data, g[pre + "M_val"])
        assert s.getNcp() == int(g[pre + "ncp"])
        assert s.getPrealloc() == int(g[pre + "prealloc"])
        assert s.getDegree() == int(g[pre + "degree"])
        assert int(s.needsDG()) == int(g[pre + "needsDG"])


def test_scalar_generate_M_matches_vectorised():
    g = _load("golden_tensor.npz")
    for name in ("2d_p2_n4", "3d_p2_n2", "2d_periodic", "2d_nonuni"):
        s, pre = _case_bspline(g, name)
        X, _ = O.fe_node_grid(s)
        M = O.generate_M([s], [X])
        assert np.array_equal(M.indptr, g[pre + "M_rowptr"])
        assert np.array_equal(M.indices, g[pre + "M_col"])
        assert np.array_equal(M.data, g[pre + "M_val"])


def test_side_dofs_and_greville():
    g = _load("golden_tensor.npz")
    for name in g["names"]:
        name = str(name)
        s, pre = _case_bspline(g, name)
        for direction in range(s.nvar):
            for side in (0, 1):
                for nl in (1, 2):
                    ref = g[pre + "side_%d_%d_%d" % (direction, side, nl)]
                    assert s.getSideDofs(direction, side, nl) == list(ref)
        P = g[pre + "P"]
        nsd = P.shape[1] - 1
        mine = np.array([[O.explicit_homogeneous_coordinate(s, nsd, I, j)
                          for j in range(nsd + 1)] for I in range(s.getNcp())])
        assert np.array_equal(mine, P)


def test_M_identities_partition_of_unity_and_nnz_formula():
    for d, p, nel in ((2, 2, 8), (2, 3, 5), (2, 4, 4), (3, 2, 4), (3, 3, 3)):
        s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
        M = O.generate_M_tensor(s)
        assert abs(M.sum(axis=1) - 1.0).max() < 4e-16 * (p + 1) ** d
        nnz1 = 2 + (nel - 1) * p + nel * (p - 1) * (p + 1)
        assert M.nnz == nnz1 ** d


def test_ptap_equals_direct_bspline_galerkin_1d():
    """K = M^T A M with the exact Q_p Lagrange matrix equals the direct B-spline Galerkin
    matrix (SURVEY.md section 8c identity) -- checks a-8, a-11 restatements together."""
    for p in (2, 3, 4):
        nel = 6
        kv = O.uniform_knots(p, 0., 1., nel)
        s = O.BSpline([p], [kv])
        M = O.generate_M_tensor(s)
        Mm, Km = O.fe_1d_matrices(s.splines[0].uniqueKnots, p)
        K = O.extract_matrix(M, Km, applyBCs=False).toarray()
        Ms = O.extract_matrix(M, Mm, applyBCs=False).toarray()
        # direct Galerkin by high-order quadrature of B-spline products
        t, w = O.gauss_legendre(p + 2)
        s1 = s.splines[0]
        Kd = np.zeros_like(K)
        Md = np.zeros_like(Ms)
        hfd = 1e-6
        for e in range(nel):
            a, b = s1.uniqueKnots[e], s1.uniqueKnots[e + 1]
            for tq, wq in zip(t, w):
                u = a + (b - a) * tq
                sp_ = s1.getKnotSpan(u)
                nodes = s1.getNodes(u)
                N = s1.basisFuncs(sp_, u)
                Md[np.ix_(nodes, nodes)] += np.outer(N, N) * wq * (b - a)
        assert abs(Md - Ms).max() < 1e-14
        # stiffness: symmetric, rows sum to zero (constants in the kernel)
        assert abs(K - K.T).max() < 1e-12
        assert abs(K.sum(axis=1)).max() < 1e-11


def test_zero_rows_columns_semantics():
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    Kd = rng.standard_normal((6, 6))
    K = sp.csr_matrix(Kd)
    Z = O.zero_rows_columns(K, [1, 4, 4], diag=7.0).toarray()
    ref = Kd.copy()
    for i in (1, 4):
        ref[i, :] = 0
        ref[:, i] = 0
        ref[i, i] = 7.0
    assert np.array_equal(Z, ref)


def test_krylov_restatements_against_direct():
    p, nel = 2, 8
    s = O.BSpline([p, p], [O.uniform_knots(p, 0., 1., nel)] * 2)
    M = O.generate_M_tensor(s)
    f = lambda x: np.sin(np.pi * x)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[f, f])
    zd = []
    for direction in (0, 1):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    K = O.extract_matrix(M, A, zd)
    rhs = O.extract_vector(M, b, zd)
    Ud, _ = O.solve_linear_system(M, K, rhs, "direct")
    Uc, itc, _ = O.cg_jacobi(K, rhs, rtol=1e-10)
    Ug, itg, _ = O.gmres_jacobi(K, rhs, rtol=1e-10)
    assert np.linalg.norm(Uc - Ud) <= 1e-8 * np.linalg.norm(Ud)
    assert np.linalg.norm(Ug - Ud) <= 1e-8 * np.linalg.norm(Ud)
    assert 0 < itc < 200 and 0 < itg < 400


def test_random_patches_generated_by_the_reference():
    """56 patches drawn by the generator of the random parity runs (tests/fuzz/fuzz_parity.py: dimension, degrees per direction,
    element counts, periodic directions, continuityDrop, non-uniform knots with random multiplicities), their extraction
    matrices and side-dof lists computed by the reference's own classes (tests/golden/make_golden_random.py): the oracle
    those runs compare with reproduces them bit for bit."""
    g = _load("golden_random.npz")
    for name in [str(n) for n in g["names"]]:
        pre = name + "/"
        degs = [int(v) for v in g[pre + "degrees"]]
        kvs = [g[pre + "kvec%d" % k] for k in range(len(degs))]
        s = O.BSpline(degs, [list(kv) for kv in kvs])
        assert s.getNcp() == int(g[pre + "ncp"]) and s.getDegree() == int(g[pre + "degree"])
        M = O.generate_M_tensor(s)
        assert np.array_equal(M.indptr, g[pre + "M_rowptr"]), name
        assert np.array_equal(M.indices, g[pre + "M_col"]), name
        assert np.array_equal(M.data, g[pre + "M_val"]), name                    # bit-exact
        for direction in range(len(degs)):
            for side in (0, 1):
                for nl in (1, 2):
                    got = np.array(s.getSideDofs(direction, side, nl), dtype=np.int64)
                    assert np.array_equal(got, g[pre + "side_%d_%d_%d" % (direction, side, nl)]), (name, direction, side, nl)
    # uniform knot vectors of the cases: the oracle's uniformKnots gives the reference's bits
    import json
    for name, meta in zip([str(n) for n in g["names"]], [json.loads(str(m)) for m in g["meta"]]):
        for k, kind in enumerate(meta["kinds"]):
            if kind != "nonuniform":
                kv = O.uniform_knots(meta["ps"][k], 0.0, 1.0, meta["nels"][k], kind == "periodic", meta["drops"][k])
                assert np.array_equal(np.asarray(kv, dtype=np.float64), g[name + "/kvec%d" % k]), (name, k)


def test_random_patches_point_evaluations():
    """getNodesAndEvals of the 56 random patches at points that are not mesh nodes (random interior points; a knot and its two
    floating-point neighbours per direction): columns in the reference's order, values bit for bit."""
    g = _load("golden_random.npz")
    n = 0
    for name in [str(x) for x in g["names"]]:
        pre = name + "/"
        degs = [int(v) for v in g[pre + "degrees"]]
        s = O.BSpline(degs, [list(g[pre + "kvec%d" % k]) for k in range(len(degs))])
        pts, ptr, cols, vals = g[pre + "ev_pts"], g[pre + "ev_ptr"], g[pre + "ev_cols"], g[pre + "ev_vals"]
        for i in range(pts.shape[0]):
            ne = s.getNodesAndEvals(pts[i])
            assert [int(e[0]) for e in ne] == cols[ptr[i]:ptr[i + 1]].tolist(), (name, i)
            assert np.array_equal(np.array([e[1] for e in ne]), vals[ptr[i]:ptr[i + 1]]), (name, i)
            n += 1
    assert n > 1000
