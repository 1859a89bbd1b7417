"""-m gpu: the half-storage product of the CG solve (csrc/tg_symgrid.hip) -- the K p of ``linearSolver.solve(MTAM, MTU, MTb)``
(tIGAr/common.py:1255-1258) for a symmetric box-stencil K on a 3-D grid of control points: upper triangle stored once, the
transposed entries scattered through LDS windows.  Checked against scipy and the CSR kernel (products), against the sliced
copy (solves), for grids whose sizes are no multiples of the patch sizes, for matrices that must be declined, and for
bit-reproducibility."""
import itertools
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from tigar_amd import device
    device.device_info()          # raises loudly if the library / GPU is missing
    return device


def _box_stencil(rng, shape, reach, symmetric=True):
    """random box-stencil matrix on an (n0, n1, n2) grid (x fastest), truncated at the boundary, general CSR"""
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape[::-1])
    rows, cols = [], []
    for off in itertools.product(*[range(-reach, reach + 1)] * 3):
        src = [slice(max(0, -o), s - max(0, o)) for o, s in zip(off[::-1], shape[::-1])]
        dst = [slice(max(0, o), s - max(0, -o)) for o, s in zip(off[::-1], shape[::-1])]
        rows.append(idx[tuple(src)].ravel())
        cols.append(idx[tuple(dst)].ravel())
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, n))
    if symmetric:
        A = (A + A.T).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("shape,reach", [((41, 33, 19), 3), ((24, 16, 8), 3), ((25, 17, 9), 2), ((49, 35, 11), 2),
                                         ((16, 50, 21), 1), ((73, 18, 30), 1), ((47, 47, 47), 3),
                                         ((33, 25, 12), 4), ((16, 16, 10), 4), ((45, 41, 23), 4)])     # (radius 4: round 6)
def test_half_storage_product_matches_scipy(dev, shape, reach):
    rng = np.random.default_rng(sum(shape) + reach)
    A = _box_stencil(rng, shape, reach)
    x = rng.standard_normal(A.shape[0])
    dA, dx = dev.DeviceCSR.from_scipy(A), dev.DeviceVector(data=x)
    y, info = dA.mult_symgrid(dx)
    assert info is not None, "a symmetric box stencil was declined"
    ref = A @ x
    scale = np.abs(A) @ np.abs(x)
    assert np.max(np.abs(y.get_local() - ref) / scale) < 1e-14
    # the diagonal and what follows it: ((2 reach + 1)^3 + 1) / 2 positions per row (padded to pairs), 8 B each
    npos = ((2 * reach + 1) ** 3 + 1) // 2
    assert info["value_bytes"] == A.shape[0] * ((npos + 1) // 2) * 16
    if min(shape) >= 40:      # (little boundary truncation: about half of the 8 B per stored entry of the sliced copy)
        assert info["value_bytes"] < 0.6 * 8 * A.nnz
    # the same bits in every run (one wave per window, LDS operations of a wave in order, windows summed in a fixed order)
    y2, _ = dA.mult_symgrid(dx)
    assert np.array_equal(y.get_local().view(np.int64), y2.get_local().view(np.int64))


@pytest.mark.parametrize("shape,reach,cuts", [((20, 18, 24), 2, (0, 8, 16, 24)), ((26, 17, 30), 3, (0, 9, 19, 30)),
                                              ((16, 33, 12), 1, (0, 4, 8, 12)), ((19, 21, 33), 4, (0, 10, 21, 33))])
def test_half_storage_product_of_a_z_slab(dev, shape, reach, cuts):
    """several ranks: a rank holds whole planes [z0, z1) of the grid (all columns) and x with the halo planes of its
    neighbours.  Its rows' entries ABOVE the slab are multiplied but not scattered (the next rank owns those rows), the
    entries BELOW the slab of its first planes come straight from the CSR arrays (the previous rank stores them as its upper
    triangle): the block product equals the rows of A x, for the first, an inner and the last slab"""
    rng = np.random.default_rng(sum(shape) * reach)
    A = _box_stencil(rng, shape, reach)
    x = rng.standard_normal(A.shape[0])
    dx = dev.DeviceVector(data=x)
    n01 = shape[0] * shape[1]
    ref_all = A @ x
    scale = np.abs(A) @ np.abs(x)
    for z0, z1 in zip(cuts[:-1], cuts[1:]):
        r0, r1 = z0 * n01, z1 * n01
        B = A[r0:r1].tocsr()
        B.sort_indices()
        dB = dev.DeviceCSR.from_scipy(B)
        y, info = dB.mult_symgrid(dx, row0=r0)
        assert info is not None, (z0, z1)
        assert np.max(np.abs(y.get_local() - ref_all[r0:r1]) / scale[r0:r1]) < 1e-14, (z0, z1)
        y2, _ = dB.mult_symgrid(dx, row0=r0)
        assert np.array_equal(y.get_local().view(np.int64), y2.get_local().view(np.int64))
    # a slab thinner than 2 reach + 2 planes keeps the sliced copy; so does a block that is not made of whole planes
    thin = A[0:(2 * reach + 1) * n01].tocsr()
    assert dev.DeviceCSR.from_scipy(thin).mult_symgrid(dx, row0=0) == (None, None)
    odd = A[n01 + 5:10 * n01 + 5].tocsr()
    assert dev.DeviceCSR.from_scipy(odd).mult_symgrid(dx, row0=n01 + 5) == (None, None)


def test_random_grids_slabs_and_chunkings(dev, monkeypatch):
    """seeded random run: stencil radius, grid sizes (patch sizes 24 x 16 and the sub-steps of 64 rows never divide them),
    z cuts into slabs and the number of z chunks per patch -- whole matrix and every slab against scipy"""
    rng = np.random.default_rng(20250929)
    for case in range(32):
        reach = int(rng.integers(1, 4)) if case < 24 else 4          # (the last eight: radius 4, round 6)
        shape = (int(rng.integers(16, 58)), int(rng.integers(16, 40)), int(rng.integers(2 * reach + 2, 26)))
        while np.prod(shape) * (2 * reach + 1) ** 3 > 9e6:
            shape = (shape[0] - 3, shape[1] - 2, shape[2])
            if shape[0] < 16 or shape[1] < 16:
                shape = (16, 16, shape[2])
                break
        monkeypatch.setenv("TIGAR_SYMGRID_CHUNKS", str(int(rng.integers(0, 7))))
        A = _box_stencil(rng, shape, reach)
        x = rng.standard_normal(A.shape[0])
        dx = dev.DeviceVector(data=x)
        ref = A @ x
        scale = np.abs(A) @ np.abs(x)
        y, info = dev.DeviceCSR.from_scipy(A).mult_symgrid(dx)
        assert info is not None, (case, shape, reach)
        assert np.max(np.abs(y.get_local() - ref) / scale) < 1e-14, (case, shape, reach)
        n01, n2 = shape[0] * shape[1], shape[2]
        if n2 >= 2 * (2 * reach + 2):
            cut = int(rng.integers(2 * reach + 2, n2 - (2 * reach + 2) + 1))
            for z0, z1 in ((0, cut), (cut, n2)):
                B = A[z0 * n01:z1 * n01].tocsr()
                B.sort_indices()
                yb, infob = dev.DeviceCSR.from_scipy(B).mult_symgrid(dx, row0=z0 * n01)
                assert infob is not None, (case, shape, reach, z0, z1)
                assert np.max(np.abs(yb.get_local() - ref[z0 * n01:z1 * n01]) / scale[z0 * n01:z1 * n01]) < 1e-14, \
                    (case, shape, reach, z0, z1)


def test_matrices_without_the_structure_are_declined(dev):
    rng = np.random.default_rng(3)
    shape = (30, 20, 12)
    x = dev.DeviceVector(data=rng.standard_normal(int(np.prod(shape))))
    # not symmetric: found by the comparison with the CSR product
    A = _box_stencil(rng, shape, 2, symmetric=False)
    assert dev.DeviceCSR.from_scipy(A).mult_symgrid(x) == (None, None)
    # symmetric, but one value off in the lower triangle (which the copy never reads)
    B = _box_stencil(rng, shape, 2)
    r = 3000
    lo = B.indptr[r]
    assert B.indices[lo] < r
    B.data[lo] += 0.5
    assert dev.DeviceCSR.from_scipy(B).mult_symgrid(x) == (None, None)
    # a row with an entry missing from the box (another row length), and one with a column outside it
    Cm = _box_stencil(rng, shape, 2).tolil()
    i = 4000
    j = Cm.rows[i][5]
    Cm[i, j] = 0.0
    Cm[j, i] = 0.0
    Cm = Cm.tocsr()
    Cm.eliminate_zeros()
    assert dev.DeviceCSR.from_scipy(Cm).mult_symgrid(x) == (None, None)
    D = _box_stencil(rng, shape, 2).tocoo()
    far = sp.coo_matrix(([1.0, 1.0], ([10, 5000], [5000, 10])), shape=D.shape)
    Dm = (D + far).tocsr()
    Dm.sort_indices()
    assert dev.DeviceCSR.from_scipy(Dm).mult_symgrid(x) == (None, None)
    # two dimensions / no grid at all
    E = sp.diags([1.0, 4.0, 1.0], [-1, 0, 1], shape=(6000, 6000)).tocsr()
    assert dev.DeviceCSR.from_scipy(E).mult_symgrid(dev.DeviceVector(6000)) == (None, None)


def _poisson3d(p, nel, mapped=False):
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    kv = [B.uniformKnots(p, 0., 1., n) for n in nel]
    if mapped:
        from geom_util import rational_volume
        from tigar_amd import NURBS as N
        kvs, C = rational_volume(p, nel)
        mesh = N.NURBSControlMesh([p] * 3, kvs, C)
    else:
        mesh = B.ExplicitBSplineControlMesh([p] * 3, kv)
    gen = t.EqualOrderSpline(1, mesh)
    s0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    if mapped:
        K = spline.assembleMatrix(F.LaplaceForm(geometry=gen))
        rhs = spline.assembleVector(F.NodalLoadForm(1.0, gen))
    else:
        K = spline.assembleMatrix(F.LaplaceForm())
        rhs = spline.assembleVector(F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * 3, scale=3 * np.pi ** 2))
    return spline, K, rhs


@pytest.mark.parametrize("p,nel,mapped", [(3, (38, 38, 38), False), (2, (45, 37, 41), False), (1, (50, 44, 40), False),
                                          (2, (40, 40, 40), True), (4, (37, 38, 39), False)])
def test_cg_solve_on_the_half_storage_copy(dev, p, nel, mapped, monkeypatch):
    """the K of the API (Dirichlet rows and columns included) is accepted; the solve agrees with the one on the sliced copy:
    same iteration count (+-1: the rows are summed in another order) and the same solution to the tolerance"""
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    spline, K, rhs = _poisson3d(p, nel, mapped)
    n = K.shape[0]
    assert n >= 65536
    x = DeviceVector(data=np.random.default_rng(1).standard_normal(n))
    y, info = K.mult_symgrid(x)
    assert info is not None
    y0 = K.mult(x).get_local()
    assert np.max(np.abs(y.get_local() - y0)) <= 1e-13 * np.max(np.abs(y0))
    res = {}
    monkeypatch.setenv("TIGAR_KSP_PERSISTENT", "0")      # (a K that fits into the registers of the chip never gets here)
    for mode in ("0", "1"):
        monkeypatch.setenv("TIGAR_SPMV_SYM", mode)
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-9
        U = DeviceVector(n)
        c0 = dev.prof_get(7)[1]
        its = ks.solve(K, U, rhs)
        assert ks.last["status"] == 0
        assert dev.prof_get(7)[1] - c0 == int(mode)
        res[mode] = (its, U.get_local())
    assert abs(res["0"][0] - res["1"][0]) <= 1, (res["0"][0], res["1"][0])
    assert np.max(np.abs(res["0"][1] - res["1"][1])) <= 1e-7 * np.max(np.abs(res["0"][1]))
    r = rhs.get_local() - K.to_scipy() @ res["1"][1]
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(rhs.get_local())
    # bit for bit the same in a second solve
    monkeypatch.setenv("TIGAR_SPMV_SYM", "1")
    ks = t.PETScKrylovSolver("cg", "jacobi")
    ks.parameters["relative_tolerance"] = 1e-9
    U2 = DeviceVector(n)
    assert ks.solve(K, U2, rhs) == res["1"][0]
    assert np.array_equal(U2.get_local().view(np.int64), res["1"][1].view(np.int64))


def test_chebyshev_cg_on_the_half_storage_copy(dev, monkeypatch):
    """CG with the Chebyshev polynomial preconditioner multiplies by the same copy (degree products per iteration)"""
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    spline, K, rhs = _poisson3d(2, (40, 40, 40))
    n = K.shape[0]
    monkeypatch.setenv("TIGAR_KSP_PERSISTENT", "0")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TIGAR_SPMV_SYM", mode)
        ks = t.PETScKrylovSolver("cg", "chebyshev")
        ks.parameters["relative_tolerance"] = 1e-9
        U = DeviceVector(n)
        c0 = dev.prof_get(7)[1]
        its = ks.solve(K, U, rhs)
        assert ks.last["status"] == 0 and dev.prof_get(7)[1] - c0 == int(mode)
        out[mode] = (its, U.get_local())
    assert abs(out["0"][0] - out["1"][0]) <= 1
    assert np.max(np.abs(out["0"][1] - out["1"][1])) <= 1e-7 * np.max(np.abs(out["0"][1]))


def test_matrices_of_symmetric_forms_skip_the_comparison_with_the_csr_product(dev, monkeypatch):
    """``assembleMatrix`` marks K when its symmetry is PROVED: the form says a(u, v) = a(v, u) for its own class (an instance
    attribute: a subclass that adds terms does not inherit it) and A is a Kronecker sum whose 1-D terms are checked
    to be symmetric.  The CG solve then passes TG_KSP_SYMMETRIC and the half-storage copy is built without the check against the
    CSR product (its rows are still checked one by one against the box stencil).  Every other matrix is compared ONCE: the
    result stays on the matrix -- a second solve does not compare again, a matrix that failed is not tried again.  The flag
    itself is the caller's word, as with KSPCG: a matrix that is NOT symmetric is accepted with it."""
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    from tigar_amd import forms as F, common as tc
    spline, K, rhs = _poisson3d(2, (40, 40, 40))
    assert K.symmetric_by_construction is True
    K2 = spline.extractMatrix(F.LaplaceForm().assemble_matrix(spline.V))          # a matrix handed over: no mark
    assert not getattr(K2, "symmetric_by_construction", False)

    class Convected(F.LaplaceForm):                                               # a subclass: the mark is not inherited
        pass
    assert F.LaplaceForm().symmetric is True and Convected().symmetric is False
    assert not getattr(spline.assembleMatrix(Convected()), "symmetric_by_construction", False)
    # the proof looks at the 1-D factors: symmetric terms, or terms that come with their transposes
    Cx = sp.random(9, 9, 0.4, random_state=1, format="csr")
    Sx = (Cx + Cx.T).tocsr()
    assert tc._kron_sum_is_symmetric([[Sx, Sx]]) and tc._kron_sum_is_symmetric([[Cx.T.tocsr(), Cx], [Cx, Cx.T.tocsr()], [Sx, Sx]])
    assert not tc._kron_sum_is_symmetric([[Cx, Sx]]) and not tc._kron_sum_is_symmetric([[Cx.T.tocsr(), Cx], [Sx, Sx]])
    assert not tc._kron_sum_is_symmetric(None)
    monkeypatch.setenv("TIGAR_KSP_PERSISTENT", "0")
    monkeypatch.setenv("TIGAR_SPMV_SYM", "2")
    rng = np.random.default_rng(4)
    A = _box_stencil(rng, (24, 20, 12), 2, symmetric=False)
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsr()
    A.sort_indices()
    b = DeviceVector(data=rng.standard_normal(A.shape[0]))
    for hint, used in ((False, 0), (True, 1)):
        dA = dev.DeviceCSR.from_scipy(A)
        c0 = dev.prof_get(7)[1]
        dev.krylov_solve(dA, b, DeviceVector(A.shape[0]), "cg", "jacobi", 1e-8, 1e-300, 20, 30, symmetric=hint)
        assert dev.prof_get(7)[1] - c0 == used
        if not hint:
            # compared once and found wanting: declined from now on, whatever the caller says
            dev.krylov_solve(dA, b, DeviceVector(A.shape[0]), "cg", "jacobi", 1e-8, 1e-300, 20, 30, symmetric=True)
            assert dev.prof_get(7)[1] - c0 == 0
    # a symmetric matrix handed in: compared at the first solve, not at the second (TIGAR_TRACE prints the comparison)
    As = _box_stencil(rng, (24, 20, 12), 2, symmetric=True)
    As = (As + sp.diags(np.asarray(abs(As).sum(axis=1)).ravel() + 1.0)).tocsr()
    As.sort_indices()
    dS = dev.DeviceCSR.from_scipy(As)
    for _ in range(2):
        c0 = dev.prof_get(7)[1]
        dev.krylov_solve(dS, b, DeviceVector(As.shape[0]), "cg", "jacobi", 1e-8, 1e-300, 20, 30)
        assert dev.prof_get(7)[1] - c0 == 1


def test_small_systems_and_other_solvers_keep_their_kernels(dev, monkeypatch):
    """below 65536 rows (the persistent kernels' range) and for gmres nothing changes; TIGAR_SPMV_SYM=2 forces the copy"""
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    rng = np.random.default_rng(9)
    A = _box_stencil(rng, (20, 18, 17), 2)
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsr()
    A.sort_indices()
    b = rng.standard_normal(A.shape[0])
    dA, db = dev.DeviceCSR.from_scipy(A), DeviceVector(data=b)
    out = {}
    for mode in ("1", "2"):
        monkeypatch.setenv("TIGAR_SPMV_SYM", mode)
        monkeypatch.setenv("TIGAR_KSP_PERSISTENT", "0")
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-10
        U = DeviceVector(A.shape[0])
        c0 = dev.prof_get(7)[1]
        ks.solve(dA, U, db)
        out[mode] = (dev.prof_get(7)[1] - c0, U.get_local())
    assert out["1"][0] == 0 and out["2"][0] == 1
    assert np.max(np.abs(out["1"][1] - out["2"][1])) <= 1e-8 * np.max(np.abs(out["1"][1]))
    assert np.linalg.norm(A @ out["2"][1] - b) <= 1e-8 * np.linalg.norm(b)


# ---- round 6: several fields on one scalar grid (K of nF x nF box-stencil blocks, symmetric as a whole) ---------------------
def _block_box_stencil(rng, shape, reach, nf, symmetric=True):
    """field-major matrix of nf x nf random box-stencil blocks on one grid (the pattern of an elasticity K)"""
    blocks = [[_box_stencil(rng, shape, reach, symmetric=False) for _ in range(nf)] for _ in range(nf)]
    A = sp.bmat(blocks, format="csr")
    if symmetric:
        A = (A + A.T).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("shape,reach,nf", [((24, 17, 12), 3, 3), ((37, 29, 16), 2, 3), ((18, 21, 9), 1, 2), ((33, 35, 21), 3, 2),
                                            ((16, 16, 8), 3, 4)])
def test_half_storage_product_of_several_fields(dev, shape, reach, nf, monkeypatch):
    """diagonal blocks stored half, an off-diagonal pair stored once (the full box of K_fg) and used for its rows and for
    the rows of K_gf: against scipy, bit-reproducible, about half the bytes of the sliced copy; a matrix whose off-diagonal
    pair is not each other's transpose is declined"""
    rng = np.random.default_rng(sum(shape) * reach + nf)
    A = _block_box_stencil(rng, shape, reach, nf)
    x = rng.standard_normal(A.shape[0])
    dA, dx = dev.DeviceCSR.from_scipy(A), dev.DeviceVector(data=x)
    y, info = dA.mult_symgrid(dx)
    assert info is not None, "a symmetric matrix of box-stencil blocks was declined"
    ref, scale = A @ x, np.abs(A) @ np.abs(x)
    assert np.max(np.abs(y.get_local() - ref) / scale) < 1e-14
    y2, _ = dA.mult_symgrid(dx)
    assert np.array_equal(y.get_local().view(np.int64), y2.get_local().view(np.int64))
    # per node: nf diagonal blocks of ((2r+1)^3 + 1) / 2 positions, nf (nf - 1) / 2 full boxes, padded to pairs, 8 B each
    s3 = (2 * reach + 1) ** 3
    assert info["value_bytes"] == int(np.prod(shape)) * 16 * (nf * (((s3 + 1) // 2 + 1) // 2) + nf * (nf - 1) // 2 * ((s3 + 1) // 2))
    assert info["value_bytes"] < 0.56 * 8 * nf * nf * s3 * int(np.prod(shape))
    for chunks in ("1", "3"):
        monkeypatch.setenv("TIGAR_SYMGRID_CHUNKS", chunks)
        y3, info3 = dA.mult_symgrid(dx)
        assert info3 is not None and np.max(np.abs(y3.get_local() - ref) / scale) < 1e-14, chunks
    monkeypatch.delenv("TIGAR_SYMGRID_CHUNKS")
    # the same matrix with the fields numbered plane by plane (dist.FieldSlabPath: k nF pd + f pd + ij)
    pd, n2 = shape[0] * shape[1], shape[2]
    f_, k_, ij_ = np.meshgrid(np.arange(nf), np.arange(n2), np.arange(pd), indexing="ij")
    new_of_old = (k_ * nf * pd + f_ * pd + ij_).ravel()
    old_of_new = np.argsort(new_of_old)
    Ap = A[old_of_new][:, old_of_new].tocsr()
    Ap.sort_indices()
    yp, infop = dev.DeviceCSR.from_scipy(Ap).mult_symgrid(dev.DeviceVector(data=x[old_of_new]))
    assert infop is not None and infop["value_bytes"] == info["value_bytes"]
    assert np.max(np.abs(yp.get_local() - ref[old_of_new]) / scale[old_of_new]) < 1e-14
    B = _block_box_stencil(rng, shape, reach, nf, symmetric=False)
    yb, infob = dev.DeviceCSR.from_scipy(B).mult_symgrid(dx)
    assert infob is None
    assert dev.DeviceCSR.from_scipy(B[old_of_new][:, old_of_new].tocsr()).mult_symgrid(dx)[1] is None
    # one entry of one off-diagonal block moved out of the box: declined as well
    C = A.tolil()
    n = int(np.prod(shape))
    C[5, n + 5 + (reach + 1)] = 1.0
    C[n + 5 + (reach + 1), 5] = 1.0
    C = C.tocsr()
    C.sort_indices()
    assert dev.DeviceCSR.from_scipy(C).mult_symgrid(dx)[1] is None


def test_cg_solve_of_three_displacement_fields_on_the_half_storage_copy(dev, monkeypatch):
    """linear elasticity on a 3-D patch (EqualOrderSpline(3, ...), tIGAr/common.py:1891-1914): the CG solve runs on the
    half-storage copy of the nine-block K and agrees with the solve on the sliced copy"""
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    from tigar_amd.device import DeviceVector
    p, nels = 2, (26, 27, 28)
    kvs = [B.uniformKnots(p, 0., 1., n) for n in nels]
    gen = t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh([p] * 3, kvs))
    sp0 = gen.getScalarSpline(0)
    for f in range(3):
        gen.addZeroDofs(f, sp0.getSideDofs(0, 0))
    spline = t.ExtractedSpline(gen, 2 * p)
    K = spline.assembleMatrix(F.ElasticityForm(1.3, 0.7))
    n = K.shape[0]
    assert n >= 65536
    rhs = DeviceVector(data=np.random.default_rng(3).standard_normal(n))
    rhs.zero_entries(np.asarray(sorted(spline.zeroDofs), dtype=np.int64), 0) if hasattr(rhs, "zero_entries") else None
    x = DeviceVector(data=np.random.default_rng(1).standard_normal(n))
    y, info = K.mult_symgrid(x)
    assert info is not None
    y0 = K.mult(x).get_local()
    assert np.max(np.abs(y.get_local() - y0)) <= 1e-13 * np.max(np.abs(y0))
    monkeypatch.setenv("TIGAR_KSP_PERSISTENT", "0")
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TIGAR_SPMV_SYM", mode)
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-9
        U = DeviceVector(n)
        c0 = dev.prof_get(7)[1]
        its = ks.solve(K, U, rhs)
        assert ks.last["status"] == 0
        assert dev.prof_get(7)[1] - c0 == int(mode)
        res[mode] = (its, U.get_local())
    assert abs(res["0"][0] - res["1"][0]) <= 1
    assert np.max(np.abs(res["0"][1] - res["1"][1])) <= 1e-7 * np.max(np.abs(res["0"][1]))
