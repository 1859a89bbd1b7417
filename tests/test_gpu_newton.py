"""Newton drivers over the path (tIGAr/common.py:1304-1348 and 504-584): same
control flow and relative-norm history as the numpy restatement ``oracle.newton_semilinear`` on
-lap u + u^3 = f (group-FE reaction term), fixed sparsity so K and M are assembled once."""
import numpy as np
import pytest

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tigar_amd
    from tigar_amd import BSplines, forms, device
    device.device_info()

    class NS:
        pass
    ns = NS()
    ns.t, ns.B, ns.F, ns.dev = tigar_amd, BSplines, forms, device
    return ns


def _setup(T, p=2, nel=8):
    B, t = T.B, T.t
    kv = [B.uniformKnots(p, 0., 1., nel)] * 2
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p, p], kv))
    sp0 = gen.getScalarSpline(0)
    for direction in (0, 1):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-13
    spline.setSolverOptions(maxIters=20, relativeTolerance=1e-9, linearSolver=solver)
    s = O.BSpline([p, p], [O.uniform_knots(p, 0., 1., nel)] * 2)
    X, _ = O.fe_node_grid(s)
    exact = np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1])
    f = 2 * np.pi ** 2 * exact + exact ** 3
    return gen, spline, s, X, exact, f


def _oracle(s, f, zd, rtol):
    Mo = O.generate_M_tensor(s)
    uk = [sp1.uniqueKnots for sp1 in s.splines]
    m1 = [O.fe_1d_matrices(u, s.splines[0].p) for u in uk]
    import scipy.sparse as sp
    Kfe = (sp.kron(m1[1][0], m1[0][1]) + sp.kron(m1[1][1], m1[0][0])).tocsr()
    Mfe = sp.kron(m1[1][0], m1[0][0]).tocsr()
    return O.newton_semilinear(Mo, Kfe, Mfe, f, lambda u: u ** 3, lambda u: 3 * u ** 2, zd, rtol=rtol)


def test_newton_loop_matches_oracle_history(T, capsys):
    gen, spline, s, X, exact, f = _setup(T)
    u = T.t.Function(spline.V)
    cube = lambda v: v.pointwise_mult(v).pointwise_mult(v)

    def dcube(v):
        w = v.pointwise_mult(v)
        w.axpy(2.0, w.copy())
        return w
    res = T.F.SemilinearResidual(u, f, cube, dcube)
    hist = spline.solveNonlinearVariationalProblem(res, res.tangent(), u)
    out = capsys.readouterr().out
    assert out.count("Solver iteration:") == len(hist) and "Relative norm" in out
    uo, Uo, ho = _oracle(s, f, list(spline.zeroDofs), 1e-9)
    assert len(hist) == len(ho)
    for a, b in zip(hist, ho):
        assert abs(a - b) <= 1e-6 * max(b, 1e-12) + 1e-12
    uh = u.vector().get_local()
    assert np.max(np.abs(uh - uo)) <= 1e-9
    assert np.max(np.abs(uh - exact)) < 5e-3                 # discretisation error at 8x8, p=2
    # passing IGA dofs: they seed u = M*dofs and come back holding the solution's dofs
    dofs = T.dev.DeviceVector(data=np.zeros(spline.M.shape[1]))
    u2 = T.t.Function(spline.V)
    res2 = T.F.SemilinearResidual(u2, f, cube, dcube)
    spline.solveNonlinearVariationalProblem(res2, res2.tangent(), u2, igaDoFs=dofs)
    assert np.max(np.abs(dofs.get_local() - Uo)) <= 1e-9
    # non-convergence raises (the reference prints and exits)
    spline.setSolverOptions(maxIters=1, relativeTolerance=1e-9, linearSolver=spline.linearSolver)
    u3 = T.t.Function(spline.V)
    res3 = T.F.SemilinearResidual(u3, f, cube, dcube)
    with pytest.raises(RuntimeError):
        spline.solveNonlinearVariationalProblem(res3, res3.tangent(), u3)


def test_extracted_nonlinear_problem_with_newton_solver(T):
    gen, spline, s, X, exact, f = _setup(T, p=3, nel=5)
    u = T.t.Function(spline.V)
    cube = lambda v: v.pointwise_mult(v).pointwise_mult(v)

    def dcube(v):
        w = v.pointwise_mult(v)
        w.axpy(2.0, w.copy())
        return w
    res = T.F.SemilinearResidual(u, f, cube, dcube)
    problem = T.t.ExtractedNonlinearProblem(spline, res, res.tangent(), u)
    newton = T.t.NewtonSolver()
    newton.parameters["relative_tolerance"] = 1e-10
    dofs = T.t.ExtractedNonlinearSolver(problem, newton).solve()
    uo, Uo, ho = _oracle(s, f, list(spline.zeroDofs), 1e-10)
    assert np.max(np.abs(u.vector().get_local() - uo)) <= 1e-8
    assert np.max(np.abs(dofs.get_local() - Uo)) <= 1e-8
    assert newton.last["iterations"] == len(ho) - 1
    # FEtoIGA is the left inverse of the prolongation on the spline space
    back = spline.FEtoIGA(u)
    assert np.max(np.abs(back.get_local() - dofs.get_local())) <= 1e-8


def test_csr_combine_and_pointwise(T):
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    A = sp.random(40, 30, density=0.2, random_state=1, format="csr")
    Bm = A.copy()
    Bm.data = rng.standard_normal(Bm.nnz)
    cs = rng.standard_normal(30)
    dA, dB = T.dev.DeviceCSR.from_scipy(A), T.dev.DeviceCSR.from_scipy(Bm)
    C = dA.combine(2.0, dB, -0.5, T.dev.DeviceVector(data=cs)).to_scipy()
    ref = 2.0 * A - 0.5 * Bm @ sp.diags(cs)
    assert abs(C - ref).max() <= 1e-15 * max(1.0, abs(ref).max())
    other = sp.random(40, 30, density=0.2, random_state=2, format="csr")
    if other.nnz == A.nnz:
        other = sp.random(40, 30, density=0.25, random_state=5, format="csr")
    with pytest.raises(T.t._lib.TigarHipError if hasattr(T.t, "_lib") else Exception):
        dA.combine(1.0, T.dev.DeviceCSR.from_scipy(other), 1.0)
    x, y = rng.standard_normal(1000), rng.standard_normal(1000)
    w = T.dev.DeviceVector(data=x).pointwise_mult(T.dev.DeviceVector(data=y)).get_local()
    assert np.array_equal(w, x * y)
