"""Pins the C + OpenMP restatement (oracle/tigar_oracle_c.c, the multi-core CPU baseline of bench.py)
against the numpy oracle, which is itself pinned against the reference's golden vectors."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O
from oracle import tigar_oracle_c as OC


@pytest.mark.parametrize("d,p,nel", [(1, 3, 6), (2, 2, 7), (2, 4, 3), (3, 2, 4), (3, 3, 3)])
def test_c_twin_matches_numpy_oracle(d, p, nel):
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    Mc = OC.generate_M_tensor(s)
    assert Mc.shape == Mo.shape
    assert np.array_equal(Mc.indptr, Mo.indptr) and np.array_equal(Mc.indices, Mo.indices)
    assert np.array_equal(Mc.data, Mo.data)                         # same order of operations: bit-exact
    Ao, bo, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.cos(3 * x)] * d)
    rng = np.random.default_rng(2)
    Ap = Ao.tocsr().copy()
    Ap.data = Ap.data * (1.0 + 0.2 * rng.standard_normal(Ap.nnz))   # no symmetry, no tensor structure
    zd = s.getSideDofs(0, 0) + s.getSideDofs(d - 1, 1)
    for A in (Ao, Ap):
        Ko = O.extract_matrix(Mo, A, zd, diag=2.5)
        Kc = OC.extract_matrix(Mo, A, zd, diag=2.5)
        assert np.array_equal(Kc.indptr, Ko.indptr) and np.array_equal(Kc.indices, Ko.indices)
        assert abs(Kc - Ko).max() <= 1e-13 * abs(Ko).max()
    yo = O.extract_vector(Mo, bo, zd)
    yc = OC.extract_vector(Mo, bo, zd)
    assert np.max(np.abs(yo - yc)) <= 1e-14 * np.max(np.abs(yo))
    Ko = O.extract_matrix(Mo, Ao, zd)
    xo, ito, ro = O.cg_jacobi(Ko, yo, rtol=1e-9)
    xc, itc, rc = OC.cg_jacobi(Ko, yo, rtol=1e-9)
    assert abs(itc - ito) <= max(3, ito // 20)       # (summation order of the OpenMP reductions differs)
    assert np.max(np.abs(xo - xc)) <= 1e-7 * np.max(np.abs(xo))     # both converged to rtol 1e-9 only
    x = rng.standard_normal(Ko.shape[1])
    assert np.max(np.abs(OC.spmv(Ko, x) - Ko @ x)) <= 1e-13 * np.max(np.abs(Ko @ x))


def test_c_twin_periodic_and_nonuniform_rows():
    kv = [O.uniform_knots(2, 0., 1., 6, True), [0, 0, 0, 0.2, 0.5, 0.55, 1, 1, 1]]
    s = O.BSpline([2, 2], kv)
    Mo, Mc = O.generate_M_tensor(s), OC.generate_M_tensor(s)
    assert np.array_equal(Mc.indptr, Mo.indptr) and np.array_equal(Mc.indices, Mo.indices) and np.array_equal(Mc.data, Mo.data)
    assert OC.num_threads() >= 1


@pytest.mark.parametrize("d,p,nel", [(2, 3, 6), (3, 2, 5), (3, 3, 3)])
def test_blocked_and_sum_factorised_products_equal_the_one_block_product(d, p, nel):
    """tgo_ptap_blocked (rows of K in blocks so that the intermediate A*M stays bounded -- what lets the CPU baseline run at
    64^3 elements) and the direction-by-direction product built from the same C Gustavson kernel give the K of the plain
    two-product PtAP: pattern identical, values to rounding; also with blocks of a single row of K."""
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    Ao, _, _, _ = O.poisson_fe_system(s)
    rng = np.random.default_rng(5)
    Ap = Ao.tocsr().copy()
    Ap.data = Ap.data * (1.0 + 0.3 * rng.standard_normal(Ap.nnz))
    zd = s.getSideDofs(0, 0) + s.getSideDofs(d - 1, 1)
    Kref = OC.extract_matrix(Mo, Ap, zd, diag=1.5)
    for cap in (1, 5000, 10 ** 9):
        Kb = OC.extract_matrix(Mo, Ap, zd, diag=1.5, max_am_entries=cap)
        assert np.array_equal(Kb.indptr, Kref.indptr) and np.array_equal(Kb.indices, Kref.indices)
        assert np.array_equal(Kb.data, Kref.data)                       # the same products in the same order, row by row
    M1 = [O.generate_M_tensor(O.BSpline([p], [O.uniform_knots(p, 0., 1., nel)])).tocsr() for _ in range(d)]
    Kf = OC.ptap_sum_factorised(M1, Ap, zd, diag=1.5, max_am_entries=20000)
    assert np.array_equal(Kf.indptr, Kref.indptr) and np.array_equal(Kf.indices, Kref.indices)
    assert abs(Kf - Kref).max() <= 1e-13 * abs(Kref).max()
