"""GPU tests of the tIGAr-compatible API shell (tigar_amd.common / BSplines / forms) against the
oracle: the demo flow of demos/poisson/poisson.py:44-122 without FEniCS."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tigar_amd
    from tigar_amd import BSplines, forms, device
    device.device_info()
    class NS: pass
    ns = NS()
    ns.t, ns.B, ns.F, ns.dev = tigar_amd, BSplines, forms, device
    return ns


def _demo(T, d, p, nel, solver):
    B, t, F = T.B, T.t, T.F
    kv = [B.uniformKnots(p, 0.0, 1.0, nel) for _ in range(d)]
    splineMesh = B.ExplicitBSplineControlMesh([p] * d, kv)
    gen = t.EqualOrderSpline(1, splineMesh)
    scalarSpline = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, scalarSpline.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    spline.setSolverOptions(linearSolver=solver)
    f = lambda x: np.sin(np.pi * x)
    a = F.LaplaceForm()
    L = F.SeparableLoadForm([f] * d, scale=d * np.pi ** 2)
    u = t.Function(spline.V)
    U = spline.solveLinearVariationalProblem(F.Equation(a, L), u)
    return gen, spline, U, u


@pytest.mark.parametrize("d,p,nel", [(2, 2, 16), (2, 2, 32), (2, 3, 10), (3, 2, 6)])      # (2, 2, 32) = BASELINE cfg1 exactly
def test_poisson_demo_flow_matches_oracle(T, d, p, nel):
    solver = T.t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-11
    gen, spline, U, u = _demo(T, d, p, nel, solver)
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    f = lambda x: np.sin(np.pi * x)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[f] * d)
    b = b * d * np.pi ** 2
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    assert list(spline.zeroDofs) == zd
    Mo = O.generate_M_tensor(s)
    M = gen.M.to_scipy()
    assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices)
    assert np.array_equal(M.data, Mo.data)
    MT = spline.MT.to_scipy()
    assert abs(MT - Mo.T).max() == 0
    Ko = O.extract_matrix(Mo, A, zd)
    rhs = O.extract_vector(Mo, b, zd)
    Uo, uo = O.solve_linear_system(Mo, Ko, rhs, "direct")
    assert np.linalg.norm(U.get_local() - Uo) <= 1e-8 * np.linalg.norm(Uo)
    assert np.linalg.norm(u.vector().get_local() - uo) <= 1e-8 * np.linalg.norm(uo)
    # control functions (a-9): cpFuncs[i] = M_control * P[:, i]; Greville geometry => x_i itself
    X, _ = O.fe_node_grid(s)
    for i in range(d):
        assert np.max(np.abs(gen.cpFuncs[i].vector().get_local() - X[:, i])) < 1e-14
    assert np.max(np.abs(gen.cpFuncs[d].vector().get_local() - 1.0)) < 1e-14
    # manufactured solution at the FE nodes
    exact = np.prod(np.sin(np.pi * X), axis=1)
    assert np.max(np.abs(u.vector().get_local() - exact)) < 0.05 * (8.0 / nel) ** (p + 1) + 1e-6


def test_default_solver_and_external_matrix_inputs(T):
    """extractMatrix / extractVector accept any FE matrix / vector on V (reef-knot style):
    scipy and numpy inputs are uploaded; linearSolver=None falls back to tight GMRES."""
    gen, spline, U, u = _demo(T, 2, 2, 8, None)
    s = O.BSpline([2, 2], [O.uniform_knots(2, 0., 1., 8)] * 2)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.sin(np.pi * x)] * 2)
    rng = np.random.default_rng(0)
    C = sp.random(A.shape[0], A.shape[0], density=0.002, random_state=rng, format="csr")
    A2 = (A + C).tocsr()                       # "contact terms added by hand"
    K = spline.extractMatrix(A2, applyBCs=True, diag=3.0).to_scipy()
    Mo = O.generate_M_tensor(s)
    Ko = O.extract_matrix(Mo, A2, list(spline.zeroDofs), diag=3.0)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    y = spline.extractVector(b, applyBCs=False).get_local()
    assert np.max(np.abs(y - Mo.T @ b)) <= 1e-12 * np.max(np.abs(b))
    y = T.t.multTranspose(spline.M, b).get_local()
    assert np.max(np.abs(y - Mo.T @ b)) <= 1e-12 * np.max(np.abs(b))
    # plan reuse (same pattern, new values) gives the same answer as a fresh plan
    K2 = spline.extractMatrix(2.0 * A2, applyBCs=False).to_scipy()
    assert abs(K2 - 2.0 * (Mo.T @ A2 @ Mo)).max() <= 1e-11 * abs(Ko).max()


def test_multi_field_and_generic_basis_fallback(T):
    B, t = T.B, T.t
    p, nel = 2, 5
    kv = [B.uniformKnots(p, 0., 1., nel)] * 2
    cm = B.ExplicitBSplineControlMesh([p, p], kv)
    gen3 = t.EqualOrderSpline(3, cm)                   # three unknown fields (shell-like)
    s = O.BSpline([p, p], [O.uniform_knots(p, 0., 1., nel)] * 2)
    Mo = O.generate_M_tensor(s, nfields=3)
    M = gen3.M.to_scipy()
    assert M.shape == Mo.shape and abs(M - Mo).max() == 0
    assert abs(gen3.MT.to_scipy() - Mo.T).max() == 0
    assert gen3.globalDof(2, 5) == 2 * s.getNcp() + 5
    gen3.addZeroDofs(1, [0, 3])
    assert gen3.zeroDofs == [s.getNcp(), s.getNcp() + 3]

    # a user-defined AbstractScalarBasis goes through the host-loop / triplet fallback (seam b-2)
    class MyBasis(t.AbstractScalarBasis):
        def __init__(self, inner):
            self.inner = inner
        def getNodesAndEvals(self, xi):
            return self.inner.getNodesAndEvals(xi)
        def getNcp(self):
            return self.inner.getNcp()
        def generateMesh(self, comm=None, degree=None, dg=False):
            return self.inner.generateMesh(degree=degree, dg=dg)
        def getDegree(self):
            return self.inner.getDegree()
        def needsDG(self):
            return False
        def useRectangularElements(self):
            return True
        def getPrealloc(self):
            return self.inner.getPrealloc()
    genF = t.FieldListSpline(cm, [MyBasis(B.BSpline([p, p], kv))])
    M1 = genF.M.to_scipy()
    Mo1 = O.generate_M_tensor(s)
    assert np.array_equal(M1.indptr, Mo1.indptr) and np.array_equal(M1.indices, Mo1.indices)
    assert np.array_equal(M1.data, Mo1.data)


def test_bspline_host_api_matches_golden(T):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_tensor.npz"))
    B = T.B
    for name in ("2d_p2_n4", "3d_p2_n2", "2d_p23_n3", "2d_periodic"):
        pre = name + "/"
        degs = [int(x) for x in g[pre + "degrees"]]
        kvecs = [g[pre + "kvec%d" % k] for k in range(len(degs))]
        s = B.BSpline(degs, kvecs)
        cm = B.ExplicitBSplineControlMesh(degs, kvecs)
        assert s.getNcp() == int(g[pre + "ncp"]) and s.getPrealloc() == int(g[pre + "prealloc"])
        assert s.getDegree() == int(g[pre + "degree"]) and int(s.needsDG()) == int(g[pre + "needsDG"])
        for direction in range(s.nvar):
            for side in (0, 1):
                for nl in (1, 2):
                    assert s.getSideDofs(direction, side, nl) == list(g[pre + "side_%d_%d_%d" % (direction, side, nl)])
        P = g[pre + "P"]
        assert np.array_equal(cm.getHomogeneousCoordinates(), P)
        assert cm.getHomogeneousCoordinate(3, 0) == P[3, 0]
        for i in range(P.shape[1]):
            assert np.array_equal(cm.homogeneousCoordinateDeviceVector(i).get_local(), P[:, i])
        # single-point API (runs the device twin): reference's entry order and values
        ne_cols, ne_vals = g[pre + "ne_cols"], g[pre + "ne_vals"]
        grid = s.generateMesh()
        X = grid.coordinates()
        for r in (0, X.shape[0] // 2, X.shape[0] - 1):
            ne = s.getNodesAndEvals(X[r])
            assert [e[0] for e in ne] == list(ne_cols[r])
            assert np.array_equal(np.array([e[1] for e in ne]), ne_vals[r])
    with pytest.raises(ValueError):
        B.uniformKnots(2, 0., 1., 4, False, 2)


def test_periodic_patch_through_the_generator_takes_the_pencil_walk(T):
    """generateM of a periodic patch through the public API: the Kronecker pencil walk (closed-form row starts, no count
    pass) now also serves periodic directions -- rows are Kronecker products of the 1-D rows sorted by function index --
    and gives the reference's matrix bit for bit, as the general count / scan / fill kernels do (TIGAR_EXTRACT_KRON=0);
    M^T b and the transposed matrix likewise."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_tensor.npz"))
    B, t = T.B, T.t
    pre = "2d_periodic/"
    degs = [int(x) for x in g[pre + "degrees"]]
    kvecs = [g[pre + "kvec%d" % k] for k in range(len(degs))]
    rp = g[pre + "M_rowptr"]
    Mg = sp.csr_matrix((g[pre + "M_val"], g[pre + "M_col"], rp), shape=(len(rp) - 1, int(g[pre + "ncp"])))
    mats = []
    for env in (None, "0"):
        if env is not None:
            os.environ["TIGAR_EXTRACT_KRON"] = env
        try:
            gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh(degs, kvecs))
        finally:
            os.environ.pop("TIGAR_EXTRACT_KRON", None)
        if env is None:
            assert gen._kron is not None and gen._kron.columns_distinct() and not gen._kron.columns_ascending()
        M = gen.M.to_scipy()
        assert np.array_equal(M.indptr, Mg.indptr) and np.array_equal(M.indices, Mg.indices)
        assert np.array_equal(M.data, Mg.data)
        MT = gen.MT.to_scipy()
        MgT = Mg.T.tocsr()
        MgT.sort_indices()
        assert np.array_equal(MT.indices, MgT.indices) and np.array_equal(MT.data, MgT.data)
        mats.append(M)
    # a 3-D patch periodic in one direction, against the oracle
    p, nel = 2, 5
    kv3 = [B.uniformKnots(p, 0., 1., nel), B.uniformKnots(p, 0., 1., nel, True), B.uniformKnots(p, 0., 1., 4)]
    gen3 = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv3))
    s3 = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., nel), O.uniform_knots(p, 0., 1., nel, periodic=True),
                             O.uniform_knots(p, 0., 1., 4)])
    Mo3 = O.generate_M_tensor(s3)
    M3 = gen3.M.to_scipy()
    assert np.array_equal(M3.indptr, Mo3.indptr) and np.array_equal(M3.indices, Mo3.indices)
    assert np.array_equal(M3.data, Mo3.data)


@pytest.mark.parametrize("nel,periodic", [(6, (0,)), (9, (0,)), (7, (2,)), (8, (0, 1, 2)), (5, (1, 2))])
def test_extract_matrix_on_a_periodic_patch(T, nel, periodic):
    """M^T A M on a patch with periodic directions (tIGAr/BSplines.py:204-212, 246-260): supports wrap around, which the
    line walks (one interval of operands per direction) cannot address -- they run on the space BEFORE the wrapped
    functions are identified (nel + p functions per periodic direction, the structure of an open knot vector) and the
    identification K = R^T K_u R follows as one pass of the general kernels (kronptap.KronExtraction.unwrapped / fold).
    Pattern and values against the oracle's product with the reference's M; the walks did run (counter)."""
    import os
    t, B, F, dev = T.t, T.B, T.F, T.dev
    d, p = 3, 2
    kv = [B.uniformKnots(p, 0., 1., nel, k in periodic) for k in range(d)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, kv))
    sp0 = gen.getScalarSpline(0)
    for direction in range(d):
        if direction not in periodic:
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    if len(periodic) == d:
        gen.addZeroDofs(0, [0, 11])                       # (all-periodic Laplacian: pin something)
    spline = t.ExtractedSpline(gen, 2 * p)
    A = F.LaplaceForm().assemble_matrix(spline.V)
    dev.prof_reset()
    K = spline.extractMatrix(A, diag=1.5).to_scipy()
    assert dev.prof_get(5)[1] > 0
    Ko = O.extract_matrix(gen.M.to_scipy(), A.to_scipy(), list(spline.zeroDofs), diag=1.5)
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # an FE matrix with a coupling outside the element pattern (added by hand): split / general stages under the fold
    A2 = A.to_scipy().tolil()
    A2[3, A2.shape[1] - 5] = 0.25
    A2 = A2.tocsr()
    K2 = spline.extractMatrix(A2, diag=1.5).to_scipy()
    Ko2 = O.extract_matrix(gen.M.to_scipy(), A2, list(spline.zeroDofs), diag=1.5)
    assert abs(K2 - Ko2).max() <= 1e-12 * abs(Ko2).max()
    K3 = spline.extractMatrix(A, diag=1.5).to_scipy()                         # bit-reproducible
    assert np.array_equal(K3.data.view(np.int64), K.data.view(np.int64))
    # the general stages alone (TIGAR_PTAP_UNWRAP=0) give the same matrix
    os.environ["TIGAR_PTAP_UNWRAP"] = "0"
    try:
        spline_g = t.ExtractedSpline(gen, 2 * p)
        dev.prof_reset()
        Kg = spline_g.extractMatrix(A, diag=1.5).to_scipy()
        assert dev.prof_get(5)[1] == 0
    finally:
        del os.environ["TIGAR_PTAP_UNWRAP"]
    assert np.array_equal(Kg.indices, K.indices) and abs(Kg - K).max() <= 1e-12 * abs(Ko).max()


@pytest.mark.parametrize("p,nel,nfields", [(2, 8, 1), (3, 9, 1), (4, 11, 1), (2, 7, 3)])
def test_extract_matrix_on_a_doubly_periodic_2d_patch(T, p, nel, nfields):
    """the 2-D line walks under periodic directions (demos/taylor-green-2d.py builds its velocity and pressure spaces
    on periodic uniform knots): one and several fields on one basis, against the oracle's product"""
    t, B, F, dev = T.t, T.B, T.F, T.dev
    kv = [B.uniformKnots(p, 0., 1., nel, True), B.uniformKnots(p, 0., 1., nel + 1, True)]
    gen = t.EqualOrderSpline(nfields, B.ExplicitBSplineControlMesh([p, p], kv))
    gen.addZeroDofs(0, [0, 5])
    spline = t.ExtractedSpline(gen, 2 * p)
    pats = []
    for n in (nel, nel + 1):                                                   # element-coupling pattern of the Q_p grid
        P1 = sp.lil_matrix((p * n + 1, p * n + 1))
        for e in range(n):
            P1[p * e:p * e + p + 1, p * e:p * e + p + 1] = 1.0
        pats.append(P1.tocsr())
    pat = O.kron_dir0_fastest(pats).tocsr()
    A = sp.bmat([[pat] * nfields for _ in range(nfields)], format="csr")
    A.sort_indices()
    A.data = np.random.default_rng(p + nel).standard_normal(A.nnz)             # values arbitrary, non-symmetric
    dev.prof_reset()
    K = spline.extractMatrix(A, diag=2.0).to_scipy()
    assert dev.prof_get(5)[1] > 0
    Ko = O.extract_matrix(gen.M.to_scipy(), A, list(spline.zeroDofs), diag=2.0)
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()


def test_slab_streaming_path_matches_resident_path(T):
    """SlabHotPath (z-slab streaming, the single-GPU form of the multi-GPU pipeline) reproduces
    the resident single-block path: K rows, M^T b, solution and prolongation."""
    from tigar_amd.dist import SlabHotPath
    from tigar_amd.common import TensorFunctionSpace
    B, F, dev = T.B, T.F, T.dev
    d, p, nel = 3, 2, 7
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    basis = B.ExplicitBSplineControlMesh([p] * d, kv).getScalarSpline()
    grid = basis.generateMesh(degree=p)
    V = TensorFunctionSpace([grid], "Lagrange")
    lap = F.LaplaceForm()
    load = F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * d, scale=2.0)
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += basis.getSideDofs(direction, side)
    ref = SlabHotPath(basis, grid)                                   # one slab = resident path
    K0, r0 = ref.assemble(lambda a, b: lap.assemble_matrix(V, a, b), lambda a, b: load.assemble_vector(V, a, b), zd)
    K0s = K0.to_scipy()
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    Ao, bo, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.sin(np.pi * x)] * d)
    Ko = O.extract_matrix(Mo, Ao, zd)
    assert abs(K0s - Ko).max() <= 1e-12 * abs(Ko).max()
    for sub in (1, 2, 4):
        path = SlabHotPath(basis, grid, sub_planes=sub)
        assert len(path.sub_slabs()) == -(-basis.splines[-1].getNcp() // sub)
        timers = {}
        K, r = path.assemble(lambda a, b: lap.assemble_matrix(V, a, b), lambda a, b: load.assemble_vector(V, a, b),
                             zd, 1.0, timers)
        Ks = K.to_scipy()
        assert np.array_equal(Ks.indptr, K0s.indptr) and np.array_equal(Ks.indices, K0s.indices)
        assert abs(Ks - K0s).max() <= 1e-13 * abs(K0s).max()
        assert np.max(np.abs(r.get_local() - r0.get_local())) <= 1e-13 * np.max(np.abs(r0.get_local()))
        U, its, res, status = path.solve(K, r, rtol=1e-10)
        assert status == 0
        u = path.prolong(U).get_local()
        Uo, uo = O.solve_linear_system(Mo, Ko, O.extract_vector(Mo, 2.0 * bo, zd), "direct")
        assert np.linalg.norm(U.get_local() - Uo) <= 1e-7 * np.linalg.norm(Uo)
        assert np.linalg.norm(u - uo) <= 1e-7 * np.linalg.norm(uo)
        assert set(timers) >= {"extract", "input", "ptap", "mtb", "stack"}


@pytest.mark.parametrize("p,nel,fused", [(2, 12, True), (3, 10, False)])
def test_periodic_patch_streamed_in_sub_slabs(T, p, nel, fused):
    """a patch periodic in x and y streamed through the slab engine (the slab direction z stays open): the tensor line walks
    run on the unwrapped space sub-slab by sub-slab, the rank's rows of K_u are folded at the end (kronptap.unwrapped /
    fold); K rows, M^T b against the one-slab path and the oracle -- FE matrix as Kronecker-sum factors fused into the
    first pass and as materialised row blocks"""
    from tigar_amd.dist import SlabHotPath
    from tigar_amd.common import TensorFunctionSpace
    B, F, dev = T.B, T.F, T.dev
    d = 3
    kv = [B.uniformKnots(p, 0., 1., nel, True), B.uniformKnots(p, 0., 1., nel + 1, True), B.uniformKnots(p, 0., 1., nel - 1)]
    basis = B.ExplicitBSplineControlMesh([p] * d, kv).getScalarSpline()
    grid = basis.generateMesh(degree=p)
    V = TensorFunctionSpace([grid], "Lagrange")
    lap = F.LaplaceForm()
    load = F.SeparableLoadForm([lambda x: np.sin(2 * np.pi * x)] * 2 + [lambda x: np.sin(np.pi * x)], scale=2.0)
    zd = basis.getSideDofs(2, 0) + basis.getSideDofs(2, 1)
    a_fac = lap.factors(V) if fused else None
    a_rows, b_rows = (lambda a, b: lap.assemble_matrix(V, a, b)), (lambda a, b: load.assemble_vector(V, a, b))
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel, True), O.uniform_knots(p, 0., 1., nel + 1, True),
                            O.uniform_knots(p, 0., 1., nel - 1)])
    Mo = O.generate_M_tensor(s)
    Ao = lap.assemble_matrix(V).to_scipy()
    Ko = O.extract_matrix(Mo, Ao, zd, diag=1.25)
    ro = O.extract_vector(Mo, load.assemble_vector(V).get_local(), zd)
    for sub in (None, 3, 5):
        path = SlabHotPath(basis, grid, sub_planes=sub)
        dev.prof_reset()
        K, r = path.assemble(a_rows, b_rows, zd, 1.25, None, a_fac)
        assert dev.prof_get(5)[1] >= len(path.sub_slabs())                   # every sub-slab through the walks
        Ks = K.to_scipy()
        assert np.array_equal(Ks.indptr, Ko.indptr) and np.array_equal(Ks.indices, Ko.indices)
        assert abs(Ks - Ko).max() <= 1e-12 * abs(Ko).max()
        assert np.max(np.abs(r.get_local() - ro)) <= 1e-12 * np.max(np.abs(ro))


def test_rccl_world1_comm_roundtrip(T):
    """RCCL communicator with one rank: slab descriptor, halo extend and all-reduce are
    identities; the distributed Krylov entry point runs through the comm branch."""
    dev = T.dev
    comm = dev.Comm(dev.Comm.unique_id(), 0, 1)
    comm.set_slab(0, 10, 0, 0, 10)
    assert comm.allreduce_sum([1.5, 2.5]).tolist() == [1.5, 2.5]
    x = dev.DeviceVector(data=np.arange(10.0))
    assert np.array_equal(comm.halo_extend(x).get_local(), np.arange(10.0))


@pytest.mark.parametrize("box", ["1", "0"])
@pytest.mark.parametrize("d,p,nel", [(2, 2, 9), (2, 4, 5), (3, 2, 5), (3, 3, 4), (3, 4, 3), (1, 3, 7)])
def test_sum_factorised_ptap_equals_direct(T, d, p, nel, box, monkeypatch):
    # box=1: dense-LDS-box Kronecker kernel (tg_ptap_kron); box=0: the general hash kernel fed
    # with explicit directional operators
    monkeypatch.setenv("TIGAR_PTAP_BOX", box)
    """K from the three directional PtAP stages == K from the one-shot PtAP == oracle, for an
    arbitrary (non-symmetric, perturbed) FE matrix; also slab by slab."""
    from tigar_amd.kronptap import KronExtraction, ptap_factored
    from tigar_amd.dist import SlabHotPath
    from tigar_amd.common import TensorFunctionSpace
    B, F, dev = T.B, T.F, T.dev
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    basis = B.ExplicitBSplineControlMesh([p] * d, kv).getScalarSpline()
    grid = basis.generateMesh(degree=p)
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    Ao, bo, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.cos(x)] * d)
    rng = np.random.default_rng(7)
    Ao = Ao.tocsr().copy()
    Ao.data = Ao.data * (1.0 + 0.3 * rng.standard_normal(Ao.nnz))        # destroys symmetry / tensor form
    kx = KronExtraction(basis, grid)
    assert kx.is_exact_for(Mo.nnz, 1e-15)
    zd = basis.getSideDofs(0, 0) + basis.getSideDofs(d - 1, 1)
    nz = grid.shape()[-1]
    ncpz = basis.splines[-1].getNcp()
    Ko = O.extract_matrix(Mo, Ao, zd, diag=2.0)
    group_sets = [None] + ([[[0, 1], [2]], [[0], [1, 2]]] if d == 3 else [])
    for groups in group_sets:
        K = ptap_factored(kx, dev.DeviceCSR.from_scipy(Ao), (0, nz), (0, nz), (0, ncpz), zd, 2.0, groups).to_scipy()
        assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
        assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # slab-streamed, factored vs direct
    V = TensorFunctionSpace([grid], "Lagrange")
    Ad = dev.DeviceCSR.from_scipy(Ao)
    def a_rows(r0, r1):
        return dev.DeviceCSR.from_scipy(Ao[r0:r1])
    def b_rows(r0, r1):
        return dev.DeviceVector(data=bo[r0:r1])
    for fac in (True, False):
        path = SlabHotPath(basis, grid, sub_planes=2, factored=fac)
        Ks, rs = path.assemble(a_rows, b_rows, zd, 2.0)
        assert abs(Ks.to_scipy() - Ko).max() <= 1e-12 * abs(Ko).max()
        assert np.max(np.abs(rs.get_local() - O.extract_vector(Mo, bo, zd))) <= 1e-12 * np.max(np.abs(bo))


def test_box_kernel_sampled_reach_falls_back_to_exact(T, monkeypatch):
    """The accumulator boxes are sized from a sampled scan of A's rows; an irregular A with a
    long-range coupling in an unsampled row must be caught (entry outside its box) and redone with
    the exact reach -- result still equals the oracle's M^T A M."""
    from tigar_amd.kronptap import KronExtraction, ptap_factored
    B, dev = T.B, T.dev
    d, p, nel = 2, 2, 9
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    basis = B.ExplicitBSplineControlMesh([p] * d, kv).getScalarSpline()
    grid = basis.generateMesh(degree=p)
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    Ao, _, _, _ = O.poisson_fe_system(s)
    Ao = Ao.tolil()
    n = Ao.shape[0]
    Ao[3, n - 5] = 0.37          # row 3 is not visited with stride 7; couples across the whole patch
    Ao[n - 2, 11] = -0.21
    Ao = Ao.tocsr()
    monkeypatch.setenv("TIGAR_BOX_REACH_STRIDE", "7")
    kx = KronExtraction(basis, grid)
    nz = grid.shape()[-1]
    K = ptap_factored(kx, dev.DeviceCSR.from_scipy(Ao), (0, nz), (0, nz), (0, basis.splines[-1].getNcp()),
                      None, 1.0, [[0], [1]])
    Ko = (Mo.T @ Ao @ Mo).tocsr()
    assert abs(K.to_scipy() - Ko).max() <= 1e-12 * abs(Ko).max()


def test_loose_row_stage_results(T):
    """Intermediate PtAP stages hand their temporary over without the row-reorder copy
    (tg_ptap_kron_stage): such loose-row matrices compact to exactly the canonical stage result,
    stack with each other, and are refused by every entry point that assumes canonical CSR."""
    from tigar_amd.kronptap import KronExtraction
    from tigar_amd import _lib
    B, F, dev = T.B, T.F, T.dev
    p, nel = 3, 5
    kv = [B.uniformKnots(p, 0., 1., nel)] * 3
    basis = B.ExplicitBSplineControlMesh([p] * 3, kv).getScalarSpline()
    grid = basis.generateMesh(degree=p)
    kx = KronExtraction(basis, grid)
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., nel)] * 3)
    Ao, _, _, _ = O.poisson_fe_system(s)
    A = dev.DeviceCSR.from_scipy(Ao)
    dims = kx.dims(set())
    fac = [kx.M1[0], None, None]
    n_out = kx.ncp[0] * kx.nfe[1] * kx.nfe[2]
    loose = dev.ptap_kron(A, 0, dims, fac, 0, n_out, intermediate=True)
    canon = dev.ptap_kron(A, 0, dims, fac, 0, n_out)
    assert loose.is_loose() and not canon.is_loose()
    C1, C2 = loose.compact().to_scipy(), canon.to_scipy()
    assert np.array_equal(C1.indptr, C2.indptr) and np.array_equal(C1.indices, C2.indices)
    assert abs(C1 - C2).max() <= 1e-13 * abs(C2).max()
    L2 = loose.to_scipy()                                    # download compacts on the fly
    assert np.array_equal(L2.indices, C2.indices)
    # two loose blocks (row halves) stack into one loose matrix equal to the whole
    half = (n_out // 2 // (kx.ncp[0] * kx.nfe[1])) * (kx.ncp[0] * kx.nfe[1])
    lo = dev.ptap_kron(A, 0, dims, fac, 0, half, intermediate=True)
    hi = dev.ptap_kron(A, 0, dims, fac, half, n_out, intermediate=True)
    st = dev.csr_vstack([lo, hi])
    assert st.is_loose()
    S = st.to_scipy()
    assert np.array_equal(S.indptr, C2.indptr) and abs(S - C2).max() <= 1e-13 * abs(C2).max()
    # the stacked VIEW (row tables only, entries stay in the blocks) compacts to the same matrix and
    # feeds the next stage like the copy does; also over canonical blocks
    vw = dev.csr_vstack_view([lo, hi])
    assert vw.is_loose()
    V = vw.to_scipy()
    assert np.array_equal(V.indptr, C2.indptr) and np.array_equal(V.indices, C2.indices) and abs(V - C2).max() <= 1e-13 * abs(C2).max()
    c_lo = dev.ptap_kron(A, 0, dims, fac, 0, half)
    c_hi = dev.ptap_kron(A, 0, dims, fac, half, n_out)
    V2 = dev.csr_vstack_view([c_lo, c_hi]).to_scipy()
    assert np.array_equal(V2.indices, C2.indices) and abs(V2 - C2).max() <= 1e-13 * abs(C2).max()
    dims_y = kx.dims({0})
    fac_y = [None, kx.M1[1], None]
    n_y = kx.ncp[0] * kx.ncp[1] * kx.nfe[2]
    y_from_view = dev.ptap_kron(vw, 0, dims_y, fac_y, 0, n_y).to_scipy()
    y_from_copy = dev.ptap_kron(st, 0, dims_y, fac_y, 0, n_y).to_scipy()
    assert np.array_equal(y_from_view.indices, y_from_copy.indices)
    assert abs(y_from_view - y_from_copy).max() <= 1e-13 * abs(y_from_copy).max()
    with pytest.raises(_lib.TigarHipError):
        dev.csr_vstack([lo, canon])
    x = dev.DeviceVector(data=np.ones(loose.shape[1]))
    with pytest.raises(_lib.TigarHipError):
        loose.mult(x)
    with pytest.raises(_lib.TigarHipError):
        loose.transpose()


def test_random_patches_generated_by_the_reference(T):
    """56 random patches whose extraction matrices and side-dof lists were computed by the REFERENCE's own classes
    (tests/golden/golden_random.npz, drawn by the generator of tests/fuzz/fuzz_parity.py): generateM of the product bit for bit
    (stored operator, its transpose, and the general count / fill kernels with TIGAR_EXTRACT_KRON=0), getSideDofs, getNcp."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_random.npz"))
    B, t = T.B, T.t
    for kron in ("1", "0"):
        os.environ["TIGAR_EXTRACT_KRON"] = kron
        try:
            for name in [str(n) for n in g["names"]]:
                pre = name + "/"
                degs = [int(x) for x in g[pre + "degrees"]]
                kvecs = [[float(v) for v in g[pre + "kvec%d" % k]] for k in range(len(degs))]
                gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh(degs, kvecs))
                M = gen.M.to_scipy()
                M.sort_indices()
                assert np.array_equal(M.indptr, g[pre + "M_rowptr"]) and np.array_equal(M.indices, g[pre + "M_col"]), name
                assert np.array_equal(M.data, g[pre + "M_val"]), name                       # bit-exact
                MT = gen.MT.to_scipy().T.tocsr()
                MT.sort_indices()
                assert np.array_equal(MT.indices, g[pre + "M_col"]) and np.array_equal(MT.data, g[pre + "M_val"]), name
                if kron == "1":
                    s = gen.getScalarSpline(0)
                    assert s.getNcp() == int(g[pre + "ncp"]) and s.getDegree() == int(g[pre + "degree"])
                    for direction in range(len(degs)):
                        for side in (0, 1):
                            for nl in (1, 2):
                                assert list(s.getSideDofs(direction, side, nl)) == \
                                    list(g[pre + "side_%d_%d_%d" % (direction, side, nl)]), (name, direction, side, nl)
        finally:
            del os.environ["TIGAR_EXTRACT_KRON"]


def test_random_patches_point_evaluations_match_the_reference(T):
    """getNodesAndEvals of the 56 random reference patches at points that are not mesh nodes (random interior points, a knot
    and its two floating-point neighbours per direction; golden_random.npz): the single-point API (columns in the
    reference's order, values bit for bit) and the explicit-coordinate extraction kernel (`tg_extract_csr_points`: the rows
    of M for dolfin-supplied node coordinates, with generateM's filter and sorted columns)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_random.npz"))
    B, dev = T.B, T.dev
    n = 0
    for name in [str(x) for x in g["names"]]:
        pre = name + "/"
        degs = [int(v) for v in g[pre + "degrees"]]
        s = B.BSpline(degs, [[float(v) for v in g[pre + "kvec%d" % k]] for k in range(len(degs))])
        pts, ptr, cols, vals = g[pre + "ev_pts"], g[pre + "ev_ptr"], g[pre + "ev_cols"], g[pre + "ev_vals"]
        for i in range(pts.shape[0]):
            ne = s.getNodesAndEvals(pts[i])
            assert [int(e[0]) for e in ne] == cols[ptr[i]:ptr[i + 1]].tolist(), (name, i)
            assert np.array_equal(np.array([e[1] for e in ne]), vals[ptr[i]:ptr[i + 1]]), (name, i)
            n += 1
        # all points at once through the points kernel: one row per point = generateM's loop body (:1566-1571)
        Mp = dev.extract_csr_points(s.splines, pts, 0, s.getNcp(), 1e-15).to_scipy()
        Mp.sort_indices()
        for i in range(pts.shape[0]):
            row = {}
            for c, v in zip(cols[ptr[i]:ptr[i + 1]], vals[ptr[i]:ptr[i + 1]]):
                if abs(v) > 1e-15:
                    row[int(c)] = float(v)
            a, b = Mp.indptr[i], Mp.indptr[i + 1]
            assert Mp.indices[a:b].tolist() == sorted(row), (name, i)
            assert np.array_equal(Mp.data[a:b], np.array([row[c] for c in sorted(row)])), (name, i)
    assert n > 1000


def test_no_device_memory_growth_over_repeated_calls(T):
    """the shell / Newton demos call the path 10^2 - 10^4 times on one ExtractedSpline: handles, plans, the
    persistent solvers' control blocks and the caching pool must not grow -- device memory in use and the pool's block count
    are the same after 10 and after 60 rounds of extractMatrix + extractVector + five solvers."""
    import gc
    t, B, F, dev = T.t, T.B, T.F, T.dev
    p, nel = 3, 32
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p, p], [B.uniformKnots(p, 0., 1., nel)] * 2))
    sp0 = gen.getScalarSpline(0)
    for k in (0, 1):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(k, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    A = dev.DeviceCSR.from_scipy((F.LaplaceForm().assemble_matrix(spline.V).to_scipy()
                                  + F.MassForm().assemble_matrix(spline.V).to_scipy()).tocsr())
    b = dev.DeviceVector(data=np.ones(A.shape[0]))

    def used():
        gc.collect()
        dev.sync()
        free, total = dev.mem_info()[:2]
        return total - free, dev.pool_stats()
    marks = []
    for it in range(61):
        K = spline.extractMatrix(A, diag=1.0)
        y = spline.extractVector(b)
        for m in (("cg", "jacobi"), ("gmres", "jacobi"), ("bicgstab", "jacobi"), ("cg", "chebyshev"), None):
            if m is None:
                spline.setSolverOptions(linearSolver=None)
            else:
                ks = t.PETScKrylovSolver(*m)
                ks.parameters["relative_tolerance"] = 1e-8
                spline.setSolverOptions(linearSolver=ks)
            u = t.Function(spline.V)
            spline.solveLinearSystem(K, y, u)
        del K, y, u
        if it in (10, 60):
            marks.append(used())
    assert marks[0] == marks[1], marks
