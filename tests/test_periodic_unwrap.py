"""CPU: the unwrapping of periodic directions (tigar_amd/kronptap.py: KronExtraction.unwrapped) -- pure host arithmetic on the
1-D node tables.  The tables come from the oracle's restatement of getNodes / basisFuncs (tIGAr/BSplines.py:310-351: functions
``(span - p + r) % ncp``), so this pins the index logic without a GPU: M_u R == M entry by entry, the unwrapped factor has the
structure the tensor line walks need (``tensorptap.local_weights`` / ``band_pattern_ok``), and K = R^T (M_u^T A M_u) R."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O


def _kx(ps, nels, periodic):
    from tigar_amd.kronptap import KronExtraction
    kx = KronExtraction.__new__(KronExtraction)
    kx.d = len(ps)
    kx.basis = kx.grid = None
    kx.M1, kx._tables = [], []
    for p, nel, per in zip(ps, nels, periodic):
        s = O.BSpline1(p, O.uniform_knots(p, 0., 1., nel, per))
        xs = O.fe_nodes_1d(s, p)
        idx = np.array([s.getNodes(x) for x in xs], dtype=np.int64)
        val = np.array([s.basisFuncs(s.getKnotSpan(x), x) for x in xs])
        rows = np.repeat(np.arange(len(xs)), p + 1)
        nz = val.ravel() != 0.0
        M1 = sp.coo_matrix((val.ravel()[nz], (rows[nz], idx.ravel()[nz])), shape=(len(xs), s.getNcp())).tocsr()
        M1.sort_indices()
        kx.M1.append(M1)
        kx._tables.append((None, idx, val))
    kx.M1T = [m.T.tocsr() for m in kx.M1]
    kx.nfe = [m.shape[0] for m in kx.M1]
    kx.ncp = [m.shape[1] for m in kx.M1]
    kx.nnz_product = int(np.prod([m.nnz for m in kx.M1], dtype=np.float64))
    return kx


@pytest.mark.parametrize("p,nel", [(1, 4), (2, 3), (2, 6), (3, 4), (3, 9), (4, 5), (4, 11)])
def test_unwrapped_factor_times_identification_is_the_periodic_factor(p, nel):
    from tigar_amd.tensorptap import local_weights, band_pattern_ok
    kx = _kx([p], [nel], [True])
    assert not kx.columns_ascending()
    ku = kx.unwrapped()
    assert ku is not None and kx.unwrapped() is ku                   # (kept on the object)
    Mu, fold = ku.M1[0], ku.fold_maps[0]
    assert Mu.shape == (p * nel + 1, nel + p) and fold.shape == (nel + p,)
    # the last p unwrapped functions are the first p again
    assert np.array_equal(fold[nel:], fold[:p]) and sorted(fold[:nel].tolist()) == list(range(nel))
    R = sp.csr_matrix((np.ones(nel + p), fold, np.arange(nel + p + 1)), shape=(nel + p, nel))
    assert abs(Mu @ R - kx.M1[0]).max() == 0.0
    # the structure of an open knot vector with simple interior knots: what the line walks are built on
    assert local_weights(Mu, p, nel) is not None and band_pattern_ok(Mu, p, nel)
    assert ku.columns_ascending() and ku.unwrapped() is None


def test_open_directions_are_left_alone():
    kx = _kx([2, 3], [5, 6], [False, True])
    ku = kx.unwrapped()
    assert ku is not None and ku.M1[0] is kx.M1[0]
    assert np.array_equal(ku.fold_maps[0], np.arange(kx.ncp[0])) and ku.ncp == [kx.ncp[0], 6 + 3]
    assert _kx([2, 2], [5, 5], [False, False]).unwrapped() is None     # nothing wraps


@pytest.mark.parametrize("ps,nels,periodic", [((2, 2), (4, 5), (True, True)), ((3, 3), (5, 4), (True, False)),
                                               ((2, 2, 2), (3, 4, 3), (False, True, True))])
def test_fold_of_the_unwrapped_product_is_the_periodic_product(ps, nels, periodic):
    kx = _kx(ps, nels, periodic)
    ku = kx.unwrapped()
    M = O.kron_dir0_fastest(kx.M1).tocsr()
    Mu = O.kron_dir0_fastest(ku.M1).tocsr()
    m = ku.fold_maps[0]
    stride = ku.wrapped_ncp[0]
    for k in range(1, kx.d):                                          # (the map fold_operators builds, direction 0 fastest)
        m = (m[None, :] + stride * ku.fold_maps[k][:, None]).ravel()
        stride *= ku.wrapped_ncp[k]
    R = sp.csr_matrix((np.ones(m.size), m, np.arange(m.size + 1)), shape=(m.size, stride))
    assert abs(Mu @ R - M).max() == 0.0
    rng = np.random.default_rng(3)
    A = sp.random(M.shape[0], M.shape[0], density=0.05, random_state=rng, format="csr") + sp.identity(M.shape[0])
    K, Ku = (M.T @ A @ M).toarray(), (Mu.T @ A @ Mu)
    assert np.max(np.abs((R.T @ Ku @ R).toarray() - K)) <= 1e-13 * np.max(np.abs(K))
