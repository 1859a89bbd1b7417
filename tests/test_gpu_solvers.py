"""-m gpu: the solvers behind the reference's seam ``self.linearSolver.solve(MTAM, MTU, MTb)`` (tIGAr/common.py:1255-1258)
beyond cg / gmres x none / jacobi: BiCGStab (KSPBCGS [ext], left preconditioning) and CG with the Chebyshev
polynomial preconditioner, plus dolfin's solver / preconditioner names, which map onto what this library has instead of
raising."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

pytestmark = pytest.mark.gpu


def _poisson(d, p, nel):
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, kv))
    s0 = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    K = spline.assembleMatrix(F.LaplaceForm())
    rhs = spline.assembleVector(F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * d, scale=d * np.pi ** 2))
    return spline, K, rhs


@pytest.mark.parametrize("d,p,nel,degree", [(2, 3, 48, 4), (2, 4, 40, 8), (3, 2, 12, 6)])
def test_cg_with_the_chebyshev_polynomial_preconditioner(d, p, nel, degree):
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    spline, K, rhs = _poisson(d, p, nel)
    Ks, b = K.to_scipy().tocsc(), rhs.get_local()
    exact = sla.spsolve(Ks, b)
    jac = t.PETScKrylovSolver("cg", "jacobi")
    che = t.PETScKrylovSolver("cg", "chebyshev")
    for s in (jac, che):
        s.parameters["relative_tolerance"] = 1e-10
    che.parameters["chebyshev_degree"] = degree
    Uj, Uc = DeviceVector(K.shape[0]), DeviceVector(K.shape[0])
    ij, ic = jac.solve(K, Uj, rhs), che.solve(K, Uc, rhs)
    assert che.last["status"] == 0
    assert np.max(np.abs(Uc.get_local() - exact)) <= 1e-7 * np.max(np.abs(exact))
    # a polynomial of degree m in D^-1 K: an outer iteration does the work of ~m Jacobi-CG iterations (CG is optimal in
    # the Krylov space, so the products do not get fewer -- the reductions and host looks do)
    assert ic <= 2.0 * ij / degree + 3, (ic, ij)
    assert ic * degree <= 2.0 * ij + 4 * degree, (ic, ij)
    # bit-reproducible, and a second solve from the solution ends at once
    U2 = DeviceVector(K.shape[0])
    che.solve(K, U2, rhs)
    assert np.array_equal(U2.get_local().view(np.int64), Uc.get_local().view(np.int64))
    che.parameters["nonzero_initial_guess"] = True
    assert che.solve(K, U2, rhs) <= 1
    # through the reference's seam, under a name dolfin users pass
    sor = t.PETScKrylovSolver("cg", "sor")
    assert sor.preconditioner == "chebyshev" and sor.preconditioner_requested == "sor" and "stands in" in sor.note
    sor.parameters["relative_tolerance"] = 1e-10
    spline.setSolverOptions(linearSolver=sor)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    assert np.max(np.abs(U.get_local() - exact)) <= 1e-7 * np.max(np.abs(exact))


def _nonsymmetric(n=40, seed=5):
    """diagonally dominant, non-symmetric, on the 2-D p=2 element-coupling pattern"""
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    kv = [B.uniformKnots(2, 0., 1., n)] * 2
    V = t.TensorFunctionSpace([B.ExplicitBSplineControlMesh([2, 2], kv).getScalarSpline().generateMesh(degree=2)], "Lagrange")
    A = F.LaplaceForm().assemble_matrix(V).to_scipy().tocsr()
    rng = np.random.default_rng(seed)
    A.data = A.data * (1.0 + 0.4 * rng.standard_normal(A.nnz))
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() * 0.25)).tocsr()
    return A, rng.standard_normal(A.shape[0])


@pytest.mark.parametrize("pc", ["jacobi", "none"])
def test_bicgstab_on_a_nonsymmetric_system(pc):
    import tigar_amd as t
    from tigar_amd.device import DeviceCSR, DeviceVector
    A, b = _nonsymmetric()
    exact = sla.spsolve(A.tocsc(), b)
    s = t.PETScKrylovSolver("bicgstab", pc)
    s.parameters["relative_tolerance"] = 1e-11
    Ad, bd, x = DeviceCSR.from_scipy(A), DeviceVector(data=b), DeviceVector(A.shape[0])
    its = s.solve(Ad, x, bd)
    assert s.last["status"] == 0 and 1 <= its < 3000
    assert np.max(np.abs(x.get_local() - exact)) <= 1e-8 * np.max(np.abs(exact))
    # the norm the solver reports is the preconditioned residual of the iterate it returns (recurred, so to rounding)
    dinv = 1.0 / A.diagonal() if pc == "jacobi" else np.ones(A.shape[0])
    true = np.linalg.norm(dinv * (b - A @ x.get_local()))
    assert abs(true - s.last["residual_norm"]) <= 1e-6 * np.linalg.norm(dinv * b) * 1e-3 + 10 * s.last["residual_norm"]
    # GMRES needs a basis for the same system; BiCGStab's iteration count is in the range of GMRES' (two products each)
    if pc == "jacobi":                       # (unpreconditioned GMRES(30) stagnates on this system: the short recurrence does not)
        g = t.PETScKrylovSolver("gmres", pc)
        g.parameters["relative_tolerance"] = 1e-11
        xg = DeviceVector(A.shape[0])
        ig = g.solve(Ad, xg, bd)
        assert its <= 6 * ig + 50, (its, ig)
    # bit-reproducible; restart from the solution; iteration limit reported; b = 0
    x2 = DeviceVector(A.shape[0])
    s.solve(Ad, x2, bd)
    assert np.array_equal(x2.get_local().view(np.int64), x.get_local().view(np.int64))
    s.parameters["nonzero_initial_guess"] = True
    assert s.solve(Ad, x2, bd) <= 1
    s.parameters["nonzero_initial_guess"] = False
    s.parameters["maximum_iterations"] = 3
    with pytest.raises(RuntimeError, match="iteration limit"):
        s.solve(Ad, DeviceVector(A.shape[0]), bd)
    s.parameters["maximum_iterations"] = 1000
    assert s.solve(Ad, x2, DeviceVector(A.shape[0])) == 0 and np.all(x2.get_local() == 0.0)


def test_dolfin_solver_and_preconditioner_names_are_mapped_not_refused():
    import tigar_amd as t
    for m, want in (("cg", "cg"), ("gmres", "gmres"), ("bicgstab", "bicgstab"), ("default", "gmres"), ("tfqmr", "bicgstab"),
                    ("minres", "gmres")):
        assert t.PETScKrylovSolver(m, "none").method == want
    for pc in ("sor", "ilu", "icc", "bjacobi", "amg", "hypre_amg"):
        s = t.PETScKrylovSolver("cg", pc)
        assert s.preconditioner == "chebyshev" and s.preconditioner_requested == pc and pc in s.note
        assert t.PETScKrylovSolver("gmres", pc).preconditioner == "jacobi"
    assert t.PETScKrylovSolver("gmres", "default").preconditioner == "jacobi"
    with pytest.raises(ValueError):
        t.PETScKrylovSolver("cg", "no-such-preconditioner")
    with pytest.raises(ValueError):
        t.PETScKrylovSolver("no-such-method")
    with pytest.raises(ValueError):
        t.PETScKrylovSolver("gmres", "chebyshev")


@pytest.mark.parametrize("d,p,nel,pc", [(2, 3, 48, "jacobi"), (2, 4, 40, "none"), (3, 2, 12, "jacobi"), (2, 2, 9, "jacobi")])
def test_persistent_cg_for_small_systems(d, p, nel, pc, monkeypatch):
    """the whole CG loop in one cooperative kernel with two device-wide barriers per iteration (csrc/tg_krylov_small.hip;
    taken by itself for systems that fit the Infinity Cache): the recurrence of the multi-kernel loop, so the same
    iteration count (sums are formed in another order: +-1), the same solution; initial guess, maxit, b = 0,
    bit-reproducibility"""
    import tigar_amd as t
    from tigar_amd.device import DeviceVector
    spline, K, rhs = _poisson(d, p, nel)
    Ks, b = K.to_scipy().tocsc(), rhs.get_local()
    exact = sla.spsolve(Ks, b)

    def run(mode, guess=None, rtol=1e-10, maxit=None, vec=rhs):
        monkeypatch.setenv("TIGAR_KSP_PERSISTENT", mode)
        s = t.PETScKrylovSolver("cg", pc)
        s.parameters["relative_tolerance"] = rtol
        if maxit is not None:
            s.parameters["maximum_iterations"] = maxit
            s.parameters["error_on_nonconvergence"] = False
        U = DeviceVector(K.shape[0]) if guess is None else DeviceVector(data=guess)
        if guess is not None:
            s.parameters["nonzero_initial_guess"] = True
        its = s.solve(K, U, vec)
        return its, U.get_local(), dict(s.last)

    from tigar_amd import device as _dev
    _dev.prof_reset()
    i1, U1, l1 = run("1")
    assert _dev.prof_get(6)[1] == 1                      # the persistent kernel did run
    i0, U0, l0 = run("0")
    assert _dev.prof_get(6)[1] == 1
    assert l1["status"] == 0 and abs(i1 - i0) <= 1, (i1, i0)
    assert np.max(np.abs(U1 - exact)) <= 1e-7 * np.max(np.abs(exact))
    assert np.max(np.abs(U1 - U0)) <= 1e-8 * np.max(np.abs(exact))
    assert abs(l1["residual_norm"] - l0["residual_norm"]) <= 0.5 * l0["residual_norm"] + 1e-300
    i1b, U1b, _ = run("1")
    assert i1b == i1 and np.array_equal(U1b.view(np.int64), U1.view(np.int64))          # bit-reproducible
    # a guess: from the solution at once; from a perturbed solution in fewer iterations, to the same answer
    ig, Ug, lg = run("1", guess=U1)
    assert ig <= 1
    pert = U1 * (1.0 + 1e-3 * np.cos(np.arange(U1.size)))
    ig1, Ug1, _ = run("1", guess=pert)
    ig0, Ug0, _ = run("0", guess=pert)
    assert abs(ig1 - ig0) <= 1
    assert np.max(np.abs(Ug1 - exact)) <= 1e-7 * np.max(np.abs(exact))
    # the iteration limit
    im, Um, lm = run("1", maxit=5)
    im0, Um0, lm0 = run("0", maxit=5)
    assert im == 5 == im0 and lm["status"] == lm0["status"] != 0
    assert np.max(np.abs(Um - Um0)) <= 1e-10 * np.max(np.abs(Um0))
    # b = 0
    zero = DeviceVector(K.shape[0])
    iz, Uz, lz = run("1", vec=zero)
    iz0, _, lz0 = run("0", vec=zero)
    assert iz == 0 == iz0 and not Uz.any() and lz["status"] == lz0["status"]


@pytest.mark.parametrize("pc,restart", [("jacobi", 30), ("jacobi", 5), ("none", 12)])
def test_persistent_gmres_for_small_systems(pc, restart, monkeypatch):
    """GMRES(m) in one cooperative kernel (csrc/tg_krylov_small.hip: rows of K in registers, the workgroup's rows of the
    basis in LDS, three device-wide barriers per inner iteration) against the multi-kernel loop on a non-symmetric system:
    iteration count (restart cycles included), solution, reported norm, guess, iteration limit, b = 0, bit-reproducibility"""
    import tigar_amd as t
    from tigar_amd.device import DeviceCSR, DeviceVector
    A, b = _nonsymmetric(n=36, seed=9)
    # (more weight on the diagonal: the short restarts must converge, not stagnate)
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() * 0.5)).tocsr()
    exact = sla.spsolve(A.tocsc(), b)
    Ad = DeviceCSR.from_scipy(A)

    def run(mode, guess=None, rtol=1e-10, maxit=None, vec=b):
        monkeypatch.setenv("TIGAR_KSP_PERSISTENT", mode)
        s = t.PETScKrylovSolver("gmres", pc)
        s.parameters["relative_tolerance"] = rtol
        s.parameters["gmres_restart"] = restart
        if maxit is not None:
            s.parameters["maximum_iterations"] = maxit
            s.parameters["error_on_nonconvergence"] = False
        x = DeviceVector(A.shape[0]) if guess is None else DeviceVector(data=guess)
        if guess is not None:
            s.parameters["nonzero_initial_guess"] = True
        its = s.solve(Ad, x, DeviceVector(data=vec))
        return its, x.get_local(), dict(s.last)

    from tigar_amd import device as _dev
    _dev.prof_reset()
    i1, x1, l1 = run("1")
    assert _dev.prof_get(6)[1] == 1                      # the persistent kernel did run
    i0, x0, l0 = run("0")
    assert _dev.prof_get(6)[1] == 1
    assert l1["status"] == 0 == l0["status"] and abs(i1 - i0) <= 1, (i1, i0)
    assert i1 > restart or restart == 30                                       # (the short restarts do cycle)
    assert np.max(np.abs(x1 - exact)) <= 1e-7 * np.max(np.abs(exact))
    assert np.max(np.abs(x1 - x0)) <= 1e-8 * np.max(np.abs(exact))
    assert abs(l1["residual_norm"] - l0["residual_norm"]) <= 0.5 * l0["residual_norm"] + 1e-300
    i1b, x1b, _ = run("1")
    assert i1b == i1 and np.array_equal(x1b.view(np.int64), x1.view(np.int64))  # bit-reproducible
    ig, xg, lg = run("1", guess=x1)
    assert ig <= 1 and lg["status"] == 0
    pert = x1 * (1.0 + 1e-3 * np.cos(np.arange(x1.size)))
    ig1, xg1, _ = run("1", guess=pert)
    ig0, xg0, _ = run("0", guess=pert)
    assert abs(ig1 - ig0) <= 1
    assert np.max(np.abs(xg1 - exact)) <= 1e-7 * np.max(np.abs(exact))
    im, xm, lm = run("1", maxit=7)
    im0, xm0, lm0 = run("0", maxit=7)
    assert im == 7 == im0 and lm["status"] == lm0["status"] != 0
    assert np.max(np.abs(xm - xm0)) <= 1e-9 * np.max(np.abs(xm0))
    iz, xz, lz = run("1", vec=np.zeros(A.shape[0]))
    iz0, _, lz0 = run("0", vec=np.zeros(A.shape[0]))
    assert iz == 0 == iz0 and not xz.any() and lz["status"] == lz0["status"]


@pytest.mark.parametrize("pc", ["jacobi", "none"])
def test_persistent_bicgstab_for_small_systems(pc, monkeypatch):
    """BiCGStab in one cooperative kernel (four device-wide barriers per iteration, both reductions folded by every
    workgroup instead of read back on the host) against the multi-kernel loop on a non-symmetric system"""
    import tigar_amd as t
    from tigar_amd.device import DeviceCSR, DeviceVector
    A, b = _nonsymmetric(n=36, seed=11)
    exact = sla.spsolve(A.tocsc(), b)
    Ad = DeviceCSR.from_scipy(A)

    def run(mode, guess=None, rtol=1e-11, maxit=None, vec=b):
        monkeypatch.setenv("TIGAR_KSP_PERSISTENT", mode)
        s = t.PETScKrylovSolver("bicgstab", pc)
        s.parameters["relative_tolerance"] = rtol
        if maxit is not None:
            s.parameters["maximum_iterations"] = maxit
            s.parameters["error_on_nonconvergence"] = False
        x = DeviceVector(A.shape[0]) if guess is None else DeviceVector(data=guess)
        if guess is not None:
            s.parameters["nonzero_initial_guess"] = True
        its = s.solve(Ad, x, DeviceVector(data=vec))
        return its, x.get_local(), dict(s.last)

    from tigar_amd import device as _dev
    _dev.prof_reset()
    i1, x1, l1 = run("1")
    assert _dev.prof_get(6)[1] == 1                      # the persistent kernel did run
    i0, x0, l0 = run("0")
    assert _dev.prof_get(6)[1] == 1
    # (BiCGStab converges erratically on this system and amplifies the last bits of the sums: 274 against 233 iterations)
    assert l1["status"] == 0 == l0["status"] and abs(i1 - i0) <= 0.35 * i0 + 10, (i1, i0)
    assert np.max(np.abs(x1 - exact)) <= 1e-8 * np.max(np.abs(exact))
    assert np.max(np.abs(x0 - exact)) <= 1e-8 * np.max(np.abs(exact))
    i1b, x1b, _ = run("1")
    assert i1b == i1 and np.array_equal(x1b.view(np.int64), x1.view(np.int64))  # bit-reproducible
    ig, xg, lg = run("1", guess=x1)
    assert ig <= 1 and lg["status"] == 0
    pert = x1 * (1.0 + 1e-3 * np.cos(np.arange(x1.size)))
    ig1, xg1, _ = run("1", guess=pert)
    assert np.max(np.abs(xg1 - exact)) <= 1e-8 * np.max(np.abs(exact))
    # the iteration limit; the first iterates of the two loops agree to rounding (later ones drift apart: on this system the
    # method multiplies a difference by 30-1000 per iteration -- 1e-16, 5e-15, 9e-13, 1e-9, 8e-6 after 1, 2, 3, 4, 6)
    for lim, eps in ((1, 1e-13), (2, 1e-12), (6, 1e-3)):
        im, xm, lm = run("1", maxit=lim)
        im0, xm0, lm0 = run("0", maxit=lim)
        assert im == lim == im0 and lm["status"] == lm0["status"] != 0
        assert np.max(np.abs(xm - xm0)) <= eps * np.max(np.abs(xm0))
    iz, xz, lz = run("1", vec=np.zeros(A.shape[0]))
    iz0, _, lz0 = run("0", vec=np.zeros(A.shape[0]))
    assert iz == 0 == iz0 and not xz.any() and lz["status"] == lz0["status"]


@pytest.mark.parametrize("d,p,nel,degree", [(2, 3, 48, 4), (2, 4, 40, 8), (3, 2, 12, 6), (2, 2, 30, 1)])
def test_persistent_chebyshev_cg_for_small_systems(d, p, nel, degree, monkeypatch):
    """CG with the Chebyshev polynomial preconditioner in one cooperative kernel (one device-wide barrier per inner product,
    two gather buffers used alternately) against the multi-kernel loop: the interval comes from the same Lanczos steps, so
    the iteration counts agree (+-1) and the solutions to the tolerance"""
    import tigar_amd as t
    from tigar_amd import device as _dev
    from tigar_amd.device import DeviceVector
    spline, K, rhs = _poisson(d, p, nel)
    exact = sla.spsolve(K.to_scipy().tocsc(), rhs.get_local())

    def run(mode, guess=None, maxit=None, vec=rhs):
        monkeypatch.setenv("TIGAR_KSP_PERSISTENT", mode)
        s = t.PETScKrylovSolver("cg", "chebyshev")
        s.parameters["relative_tolerance"] = 1e-10
        s.parameters["chebyshev_degree"] = degree
        if maxit is not None:
            s.parameters["maximum_iterations"] = maxit
            s.parameters["error_on_nonconvergence"] = False
        U = DeviceVector(K.shape[0]) if guess is None else DeviceVector(data=guess)
        if guess is not None:
            s.parameters["nonzero_initial_guess"] = True
        its = s.solve(K, U, vec)
        return its, U.get_local(), dict(s.last)

    _dev.prof_reset()
    i1, U1, l1 = run("1")
    assert _dev.prof_get(6)[1] == 1                      # the persistent kernel did run
    i0, U0, l0 = run("0")
    assert _dev.prof_get(6)[1] == 1
    assert l1["status"] == 0 == l0["status"] and abs(i1 - i0) <= 1, (i1, i0)
    assert np.max(np.abs(U1 - exact)) <= 1e-7 * np.max(np.abs(exact))
    assert np.max(np.abs(U1 - U0)) <= 1e-8 * np.max(np.abs(exact))
    i1b, U1b, _ = run("1")
    assert i1b == i1 and np.array_equal(U1b.view(np.int64), U1.view(np.int64))
    ig, _, lg = run("1", guess=U1)
    assert ig <= 1 and lg["status"] == 0
    im, Um, lm = run("1", maxit=3)
    im0, Um0, lm0 = run("0", maxit=3)
    assert im == 3 == im0 and lm["status"] == lm0["status"] != 0
    assert np.max(np.abs(Um - Um0)) <= 1e-10 * np.max(np.abs(Um0))


def test_chebyshev_cg_with_a_right_hand_side_that_is_one_eigenvector():
    """found by the random multi-rank runs: on a uniform degree-1 patch the sine load of the demos is an eigenvector of
    D^-1 K, the Lanczos estimate started from it ended after one step with the upper end of the spectrum unseen, and the
    Chebyshev polynomial on an interval that ends below lambda_max amplified rounding noise until the recurrence broke down
    (status -2).  The start vector is modulated by a hash of the row index now (csrc/tg_krylov.hip: k_lz_start)."""
    import scipy.sparse as sp
    from tigar_amd import device as dev
    n = 24
    T = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    I = sp.identity(n)
    K = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr()
    K.sort_indices()
    x1 = np.sin(np.pi * np.arange(1, n + 1) / (n + 1))
    b = np.kron(np.kron(x1, x1), x1)                                    # the lowest eigenvector of K (constant diagonal)
    Kd = dev.DeviceCSR.from_scipy(K)
    for degree in (2, 4, 8, 16):
        x = dev.DeviceVector(K.shape[0])
        its, res, status = dev.krylov_solve(Kd, dev.DeviceVector(data=b), x, "cg", "chebyshev", 1e-10, 1e-300, 500, degree)
        assert status == 0 and its <= 3
        assert np.linalg.norm(K @ x.get_local() - b) <= 1e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("n1", [12, 40])        # 144 unknowns: the multi-kernel loops; 1 600: the persistent kernels
def test_nan_in_the_matrix_or_the_right_hand_side_is_a_breakdown_for_every_solver(n1):
    """a NaN in K (an FE matrix with a NaN entry goes through M^T A M entry by entry, as PETSc's product would) or in b ends
    every Krylov solver with status -2 (dolfin: RuntimeError) within a few iterations -- found by probing: Chebyshev-CG took
    the NaN norm of its Lanczos start vector for "b = 0" and returned x = 0 as converged; BiCGStab's max(0, ||r||^2) turned a
    NaN into 0 = converged (an Inf in K)."""
    from tigar_amd import device as dev
    T1 = sp.diags([-np.ones(n1 - 1), 2.0 * np.ones(n1), -np.ones(n1 - 1)], [-1, 0, 1])
    K = (sp.kron(T1, sp.identity(n1)) + sp.kron(sp.identity(n1), T1)).tocsr()
    K.sort_indices()
    b = np.ones(K.shape[0])
    Kbad = K.copy()
    Kbad.data[Kbad.nnz // 2] = np.nan
    bbad = b.copy()
    bbad[7] = np.nan
    Kinf = K.copy()
    Kinf.data[5] = np.inf
    for Km, bm in ((Kbad, b), (K, bbad), (Kinf, b)):
        Kd = dev.DeviceCSR.from_scipy(Km)
        for method, pc in (("cg", "jacobi"), ("cg", "none"), ("cg", "chebyshev"), ("gmres", "jacobi"), ("bicgstab", "jacobi")):
            x = dev.DeviceVector(K.shape[0])
            its, res, status = dev.krylov_solve(Kd, dev.DeviceVector(data=bm), x, method, pc, 1e-10, 1e-300, 500, 30)
            assert status == -2, (method, pc, status, its)
            assert its <= 35, (method, pc, its)
