"""BASELINE cfg3 at its FULL size (3-D, 256^3 elements, p=3: 454.8 M FE rows, 17.4 M DoFs, nnz(K) = 5.84e9
> 2^31) through the SAME API path bench.py times (EqualOrderSpline with an implicit M -> ExtractedSpline.
assembleMatrix / extractVector / solveLinearSystem), checked against the closed-form Kronecker oracle of
SURVEY.md section 8c:

    K = M^T A M = sum_d (x)_k [ k1_k if k == d else m1_k ],   k1 = M1^T Kfe1 M1,  m1 = M1^T Mfe1 M1

computed on the host from the ORACLE's 1-D matrices (rows of a Kronecker product need only 1-D rows), then
MatZeroRowsColumns.  Sampled rows: patch corner / boundary (BC rows), next to the boundary, interior, the
last rows, and rows whose entries lie beyond entry index 2^31 (int64 row pointers, pool-piece addressing of
the sliced copy, sub-slab seams of the streamed assembly).  M^T b is compared in full."""
import numpy as np
import pytest

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu

D, P, NEL = 3, 3, 256


def _oracle_1d():
    s1 = O.BSpline([P], [O.uniform_knots(P, 0., 1., NEL)])
    M1 = O.generate_M_tensor(s1).tocsr()                       # 769 x 259, the reference's 1-D extraction rows
    uk = s1.splines[0].uniqueKnots
    Mfe, Kfe = O.fe_1d_matrices(uk, P)
    k1 = (M1.T @ Kfe @ M1).tocsr()
    m1 = (M1.T @ Mfe @ M1).tocsr()
    k1.sort_indices()
    m1.sort_indices()
    b1 = M1.T @ O.fe_1d_load(uk, P, lambda x: np.sin(np.pi * x))
    return M1, k1, m1, b1, s1


def _oracle_row(r, n, k1, m1, zmask, diag):
    """Row r of zeroRowsColumns(sum_d kron(...)) as (cols, vals), columns ascending (x fastest)."""
    i, j, k = r % n, (r // n) % n, r // (n * n)
    rows = {}
    for name, mat in (("k", k1), ("m", m1)):
        for ax, idx in (("x", i), ("y", j), ("z", k)):
            sl = slice(mat.indptr[idx], mat.indptr[idx + 1])
            rows[name + ax] = (mat.indices[sl].astype(np.int64), mat.data[sl])
    # the 1-D patterns of k1 and m1 coincide (band of half-width p), so all three terms share columns
    cx, cy, cz = rows["kx"][0], rows["ky"][0], rows["kz"][0]
    assert np.array_equal(cx, rows["mx"][0]) and np.array_equal(cy, rows["my"][0]) and np.array_equal(cz, rows["mz"][0])
    cols = (cx[None, None, :] + n * cy[None, :, None] + n * n * cz[:, None, None]).reshape(-1)

    def outer(az, ay, ax):
        return (az[:, None, None] * ay[None, :, None] * ax[None, None, :]).reshape(-1)
    vals = outer(rows["mz"][1], rows["my"][1], rows["kx"][1]) + outer(rows["mz"][1], rows["ky"][1], rows["mx"][1]) \
        + outer(rows["kz"][1], rows["my"][1], rows["mx"][1])
    if zmask[r]:
        vals = np.where(cols == r, diag, 0.0)
    else:
        vals = np.where(zmask[cols], 0.0, vals)
    return cols, vals


def test_cfg3_full_size_through_the_api_against_kronecker_oracle():
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev
    free_b, total_b = dev.mem_info()
    if total_b < 200e9:
        pytest.skip("needs an MI355X-class HBM (K alone is 70 GB)")
    n = NEL + P
    nfe1 = NEL * P + 1
    ncp = n ** D
    nnzK1 = (NEL + P) * (2 * P + 1) - P * (P + 1)
    kv = [B.uniformKnots(P, 0., 1., NEL) for _ in range(D)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([P] * D, kv))
    assert getattr(gen.M, "is_implicit", False), "cfg3's M (271 GB) must stay implicit"
    assert gen.M.shape == (nfe1 ** D, ncp) and gen.M.nnz == 22306693375
    sp0 = gen.getScalarSpline(0)
    for direction in range(D):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * P)
    diag = 2.5
    K = spline.assembleMatrix(F.LaplaceForm(), diag=diag)        # streamed z-sub-slabs, rows > 2^31 entries
    assert K.shape == (ncp, ncp)
    assert K.nnz == nnzK1 ** D == 5841725401
    f1 = lambda x: np.sin(np.pi * x)
    load = F.SeparableLoadForm([f1] * D, scale=1.0)
    rhs = spline.assembleVector(load)

    M1, k1, m1, b1, s1 = _oracle_1d()
    zmask = np.zeros(ncp, dtype=bool)
    zmask[np.asarray(spline.zeroDofs, dtype=np.int64)] = True
    assert int(zmask.sum()) == ncp - (n - 2) ** D

    # ---- sampled rows of K: pattern identical, values to 1e-12 of the row's largest entry
    def rid(i, j, k):
        return i + n * (j + n * k)
    sub = spline._slab.sub_planes
    rows = [rid(0, 0, 0), rid(1, 0, 0), rid(1, 1, 1), rid(2, 1, 1), rid(3, 3, 3), rid(5, 4, 3), rid(128, 1, 2),
            rid(n - 2, n - 2, n - 2), rid(n - 1, n - 1, n - 1), rid(n - 2, 1, n - 2), rid(130, 131, 129),
            rid(7, 200, 100), rid(250, 3, 255), rid(n - 3, n - 4, n - 3)]
    # seams of the streamed assembly: last plane of one sub-slab / first plane of the next
    for seam in (sub, 2 * sub, (n // sub) * sub):
        if 0 < seam < n:
            rows += [rid(17, 40, seam - 1), rid(17, 40, seam)]
    # rows stored beyond entry 2^31 (and beyond 2^32): row pointer of the first sampled row must say so
    far = [rid(3, 5, 120), rid(100, 100, 140), rid(60, 255, 200), rid(n - 5, 9, n - 3)]
    assert K.rowptr_at(far[0]) > 2 ** 31 and K.rowptr_at(far[-1]) > 2 ** 32
    rows += far
    rng = np.random.default_rng(7)
    rows += [int(v) for v in rng.integers(0, ncp, 24)]
    worst = 0.0
    for r in rows:
        blk = K.rows_to_scipy(r, r + 1)
        cols, vals = _oracle_row(r, n, k1, m1, zmask, diag)
        assert np.array_equal(blk.indices.astype(np.int64), cols), "pattern of row %d" % r
        scale = max(np.max(np.abs(vals)), 1e-300)
        err = np.max(np.abs(blk.data - vals)) / scale
        worst = max(worst, err)
        assert err <= 1e-12, "row %d: %g" % (r, err)
        if zmask[r]:
            assert np.array_equal(blk.data, vals)                # BC rows are exact: diag and zeros
    # a contiguous block of rows in one download (row-range path of tg_csr_download_rows)
    r0 = rid(0, 77, 133)
    blk = K.rows_to_scipy(r0, r0 + n)
    for q in (0, 1, n // 2, n - 1):
        cols, vals = _oracle_row(r0 + q, n, k1, m1, zmask, diag)
        sl = slice(blk.indptr[q], blk.indptr[q + 1])
        assert np.array_equal(blk.indices[sl].astype(np.int64), cols)
        assert np.max(np.abs(blk.data[sl] - vals)) <= 1e-12 * max(np.max(np.abs(vals)), 1e-300)

    # ---- M^T b in full: (M1^T b1) (x) (M1^T b1) (x) (M1^T b1), zero at the boundary dofs
    ref = np.kron(b1, np.kron(b1, b1))
    ref[zmask] = 0.0
    got = rhs.get_local()
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))
    assert np.all(got[zmask] == 0.0)

    # ---- the products of the Krylov solve: sliced, pattern-compressed copy vs the CSR kernel, and both vs the
    #      oracle rows
    x = rng.standard_normal(ncp)
    dx = dev.DeviceVector(data=x)
    y_csr = K.mult(dx).get_local()
    ncls, padded = K.spmv_sell(True)
    assert ncls > 0 and K.nnz <= padded <= 1.1 * K.nnz
    y_sell = K.mult(dx).get_local()
    K.spmv_sell(False)
    scale = np.max(np.abs(y_csr))
    assert np.max(np.abs(y_sell - y_csr)) <= 1e-13 * scale
    # the half-storage copy the CG solve multiplies with (csrc/tg_symgrid.hip: diagonal + upper triangle, transposed entries
    # scattered through LDS windows): accepted at this size, half the bytes, all 17.4 M entries of the product
    y_sym, info = K.mult_symgrid(dx)
    assert info is not None and info["value_bytes"] == 16 * 86 * ncp and info["value_bytes"] < 0.52 * 8 * K.nnz
    y_sym = y_sym.get_local()
    assert np.max(np.abs(y_sym - y_csr)) <= 1e-13 * scale
    for r in rows[:20] + far:
        cols, vals = _oracle_row(r, n, k1, m1, zmask, diag)
        yr = float(vals @ x[cols])
        bound = 4e-16 * np.sqrt(len(cols)) * float(np.abs(vals) @ np.abs(x[cols])) + 1e-300
        assert abs(y_csr[r] - yr) <= 50 * bound and abs(y_sell[r] - yr) <= 50 * bound and abs(y_sym[r] - yr) <= 50 * bound

    # ---- whole-matrix checks (every one of the 5.8e9 entries takes part):
    # (a) K x for a separable x = xz (x) xy (x) xx that vanishes on the boundary: zeroRowsColumns(K0) x = P (K0 x), and
    #     K0 x = sum_d (x)_k [k1 or m1] x_k needs only 1-D products -- compared in all 17.4 M entries
    x1 = [rng.standard_normal(n) for _ in range(D)]
    for v in x1:
        v[0] = v[-1] = 0.0
    kx = [k1 @ v for v in x1]
    mx = [m1 @ v for v in x1]

    def kron3(az, ay, ax):
        return (az[:, None, None] * ay[None, :, None] * ax[None, None, :]).reshape(-1)
    ref = kron3(mx[2], mx[1], kx[0]) + kron3(mx[2], kx[1], mx[0]) + kron3(kx[2], mx[1], mx[0])
    ref[zmask] = 0.0
    xs = kron3(x1[2], x1[1], x1[0])
    got = K.mult(dev.DeviceVector(data=xs)).get_local()
    assert np.max(np.abs(got - ref)) <= 2e-12 * np.max(np.abs(ref))
    assert np.all(got[zmask] == 0.0)
    # (b) x supported on the boundary dofs only: every entry of a zeroed row or column must be exactly zero and the
    #     diagonal exactly `diag`:  K x == diag x, bit for bit
    xb = np.where(zmask, rng.standard_normal(ncp), 0.0)
    got = K.mult(dev.DeviceVector(data=xb)).get_local()
    assert np.array_equal(got, diag * xb)
    # (c) symmetry of the assembled operator: x^T K y == y^T K x for random (non-separable) x, y
    yv = rng.standard_normal(ncp)
    Kx, Ky = y_csr, K.mult(dev.DeviceVector(data=yv)).get_local()
    a, b = float(yv @ Kx), float(x @ Ky)
    assert abs(a - b) <= 1e-11 * (np.abs(yv) @ np.abs(Kx))
    # (d) without boundary conditions the Laplace operator annihilates constants (partition of unity): K0 1 = 0
    K0 = spline.assembleMatrix(F.LaplaceForm(), applyBCs=False)
    assert K0.nnz == K.nnz
    one = K0.mult(dev.DeviceVector(data=np.ones(ncp))).get_local()
    rowsum_abs = float(np.max(np.abs(k1).sum(axis=1))) * float(np.max(np.abs(m1).sum(axis=1))) ** 2 * 3.0
    assert np.max(np.abs(one)) <= 1e-12 * rowsum_abs
    del K0

    # ---- Jacobi-CG on the full system + matrix-free prolongation: manufactured solution
    load3 = F.SeparableLoadForm([f1] * D, scale=D * np.pi ** 2)
    rhs3 = spline.assembleVector(load3)
    solver = t.PETScKrylovSolver("cg", "jacobi")
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    # (diag only scales the decoupled boundary rows; their right-hand side is zero)
    U = spline.solveLinearSystem(K, rhs3, u)
    assert solver.last["status"] == 0 and 60 <= solver.last["iterations"] <= 110
    uh = u.vector().get_local()
    g = spline.V.grids[0]
    idx = np.arange(0, nfe1 ** D, 97)
    exact = np.ones(idx.size)
    stride = 1
    for k in range(D):
        exact *= np.sin(np.pi * g.axes[k][(idx // stride) % nfe1])
        stride *= nfe1
    assert np.max(np.abs(uh[idx] - exact)) < 5e-8            # O(h^4) + CG tolerance 1e-6
    # u = M U row by row on a sample: explicit rows of M (extraction kernel on a row range) times U
    Uh = U.get_local()
    r0 = 3 * nfe1 * nfe1 + 5 * nfe1
    Mrows = gen.M.rows_to_scipy(r0, r0 + 2 * nfe1)
    assert np.max(np.abs(Mrows @ Uh - uh[r0:r0 + 2 * nfe1])) <= 1e-13 * np.max(np.abs(uh))
