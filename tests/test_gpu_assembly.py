"""GPU parity tests of the FE-side operator assembly on mapped tensor patches (SURVEY.md 8f-1:
dolfin.assemble stand-in with the spline's metric-based dx and grad, tIGAr/common.py:917-945,
1206-1220) against the numpy element-loop restatement ``oracle.mapped_fe_system``."""
import numpy as np
import pytest

from oracle import tigar_oracle as O
from geom_util import quarter_annulus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tigar_amd
    from tigar_amd import BSplines, forms, device, NURBS
    device.device_info()

    class NS:
        pass
    ns = NS()
    ns.t, ns.B, ns.F, ns.dev, ns.N = tigar_amd, BSplines, forms, device, NURBS
    return ns


def _close(A, Ao, tol=1e-12):
    A = A.to_scipy()
    assert A.shape == Ao.shape
    assert abs(A - Ao).max() <= tol * abs(Ao).max()
    # pattern: element coupling, entries that vanish only numerically stay structural
    assert A.nnz >= Ao.nnz


@pytest.mark.parametrize("d,p,nel", [(1, 3, 5), (2, 2, 5), (2, 3, 3), (3, 2, 3), (2, 5, 2)])
def test_identity_geometry_equals_kronecker_forms(T, d, p, nel):
    B, t, F = T.B, T.t, T.F
    kv = [B.uniformKnots(p, 0., 1. + 0.5 * k, nel + k) for k in range(d)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, kv))
    for form_k, form_m in ((F.LaplaceForm(), F.LaplaceForm(geometry=gen)), (F.MassForm(), F.MassForm(geometry=gen))):
        Ak = form_k.assemble_matrix(gen.V).to_scipy()
        Am = form_m.assemble_matrix(gen.V).to_scipy()
        assert np.array_equal(Ak.indptr, Am.indptr) and np.array_equal(Ak.indices, Am.indices)
        assert abs(Ak - Am).max() <= 2e-12 * abs(Ak).max()


def _annulus_generator(T, nel, nfields=1):
    kv, Pf = quarter_annulus(nel)
    cm = T.N.NURBSControlMesh([2, 2], [kv, kv], Pf)
    return T.t.EqualOrderSpline(nfields, cm), kv


def test_nurbs_annulus_matches_oracle(T):
    gen, kv = _annulus_generator(T, 5)
    s = O.BSpline([2, 2], [kv, kv])
    uks = [sp1.uniqueKnots for sp1 in s.splines]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    X = np.stack([cp[0] / cp[2], cp[1] / cp[2]], axis=1)
    fn = np.sin(X[:, 0]) * np.exp(X[:, 1])
    Mo, Ko, bo = O.mapped_fe_system(uks, 2, cp, fnodal=fn)
    _close(T.F.MassForm(geometry=gen).assemble_matrix(gen.V), Mo)
    _close(T.F.LaplaceForm(geometry=gen).assemble_matrix(gen.V), Ko)
    b = T.F.NodalLoadForm(lambda x: np.sin(x[:, 0]) * np.exp(x[:, 1]), gen).assemble_vector(gen.V).get_local()
    assert np.max(np.abs(b - bo)) <= 1e-13 * np.max(np.abs(bo))
    # area of the quarter annulus = 3 pi / 4 (rational geometry is exact, quadrature is not)
    area = float(np.ones(len(fn)) @ (Mo @ np.ones(len(fn))))
    assert abs(area - 0.75 * np.pi) < 1e-6
    # more Gauss points change the matrices only at quadrature-error level, and agree with the oracle
    A4 = T.dev.assemble_mapped_matrix(uks, 2, [f.vector() for f in gen.cpFuncs], "laplace", nq=4)
    _, Ko4, _ = O.mapped_fe_system(uks, 2, cp, nq=4)
    _close(A4, Ko4)


def test_surface_and_volume_maps_match_oracle(T):
    B, t, dev = T.B, T.t, T.dev
    # surface in 3-D (d=2, nsd=3): polynomial graph z = x^2 + y over a stretched grid -> Laplace-Beltrami
    p = 2
    kv = [B.uniformKnots(p, 0., 1., 4), B.uniformKnots(p, 0., 2., 3)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p, p], kv))
    g = gen.V.grids[0]
    x = gen.cpFuncs[0].vector().get_local()
    y = gen.cpFuncs[1].vector().get_local()
    cp = [x, y, x * x + y, np.ones_like(x)]
    uks = [np.asarray(g.vertices[k]) for k in range(2)]
    Mo, Ko, _ = O.mapped_fe_system(uks, p, cp)
    dcp = [dev.DeviceVector(data=c) for c in cp]
    _close(dev.assemble_mapped_matrix(uks, p, dcp, "mass"), Mo)
    _close(dev.assemble_mapped_matrix(uks, p, dcp, "laplace"), Ko)
    # volume (d=3): smooth non-affine map with a rational weight
    kv3 = [B.uniformKnots(p, 0., 1., 2)] * 3
    gen3 = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv3))
    g3 = gen3.V.grids[0]
    X = [gen3.cpFuncs[i].vector().get_local() for i in range(3)]
    wgt = 1.0 + 0.2 * X[0] * X[1]
    cp3 = [(X[0] + 0.1 * X[1] * X[2]) * wgt, (X[1] + 0.2 * X[0] ** 2) * wgt, (X[2] * (1.0 + 0.3 * X[0])) * wgt, wgt]
    uks3 = [np.asarray(g3.vertices[k]) for k in range(3)]
    Mo3, Ko3, bo3 = O.mapped_fe_system(uks3, p, cp3, fnodal=X[0] + 2 * X[2])
    dcp3 = [dev.DeviceVector(data=c) for c in cp3]
    _close(dev.assemble_mapped_matrix(uks3, p, dcp3, "mass"), Mo3)
    _close(dev.assemble_mapped_matrix(uks3, p, dcp3, "laplace"), Ko3)
    b3 = dev.assemble_mapped_load(uks3, p, dcp3, dev.DeviceVector(data=X[0] + 2 * X[2])).get_local()
    assert np.max(np.abs(b3 - bo3)) <= 1e-13 * np.max(np.abs(bo3))


def test_poisson_on_nurbs_annulus_converges(T):
    """demos/poisson/poisson-nurbs.py flow without FEniCS: u = (r-1)(2-r) sin(2 theta), zero on
    the whole boundary of the quarter annulus; error drops at the optimal rate under refinement."""
    t, F = T.t, T.F

    def exact(x):
        r, th = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0])
        return (r - 1.0) * (2.0 - r) * np.sin(2.0 * th)

    def rhs(x):   # -(u_rr + u_r/r + u_thth/r^2)
        r, th = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0])
        return -(-2.0 + (3.0 - 2.0 * r) / r - 4.0 * (r - 1.0) * (2.0 - r) / r ** 2) * np.sin(2.0 * th)

    errs = []
    for nel in (4, 8, 16):
        gen, kv = _annulus_generator(T, nel)
        sp0 = gen.getScalarSpline(0)
        for direction in (0, 1):
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
        spline = t.ExtractedSpline(gen, 4)
        solver = t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-12
        spline.setSolverOptions(linearSolver=solver)
        u = t.Function(spline.V)
        spline.solveLinearVariationalProblem(
            F.Equation(F.LaplaceForm(geometry=gen), F.NodalLoadForm(rhs, gen)), u)
        cp = [f.vector().get_local() for f in gen.cpFuncs]
        X = np.stack([cp[0] / cp[2], cp[1] / cp[2]], axis=1)
        errs.append(np.max(np.abs(u.vector().get_local() - exact(X))))
    assert errs[1] < errs[0] / 5.0 and errs[2] < errs[1] / 5.0      # p = 2: rate ~3
    assert errs[2] < 2e-4


def test_mapped_assembly_is_bit_reproducible(T):
    """Elements are assembled colour by colour (parity of the element index per direction: elements of one colour share
    no node), without atomics: the same bits in every run, also with many elements around every node (odd element
    counts: the last colour is smaller)."""
    t, B, dev = T.t, T.B, T.dev
    for d, p, nel in ((2, 3, (37, 24)), (3, 2, (9, 7, 8))):
        gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, [B.uniformKnots(p, 0., 1., n) for n in nel]))
        g = gen.V.grids[0]
        X = [gen.cpFuncs[i].vector().get_local() for i in range(d)]
        wgt = 1.0 + 0.2 * X[0] * X[1]
        cp = [(X[i] + 0.1 * X[(i + 1) % d] ** 2) * wgt for i in range(d)] + [wgt]
        uks = [np.asarray(g.vertices[k]) for k in range(d)]
        dcp = [dev.DeviceVector(data=c) for c in cp]
        f = dev.DeviceVector(data=np.sin(3 * X[0]) + X[1])
        Mo, Ko, bo = O.mapped_fe_system(uks, p, cp, fnodal=f.get_local()) if np.prod(nel) < 600 else (None, None, None)
        runs = [(dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy(), dev.assemble_mapped_load(uks, p, dcp, f).get_local())
                for _ in range(3)]
        for K, b in runs[1:]:
            assert np.array_equal(K.data.view(np.int64), runs[0][0].data.view(np.int64))
            assert np.array_equal(b.view(np.int64), runs[0][1].view(np.int64))
        if Ko is not None:
            assert abs(runs[0][0] - Ko).max() <= 1e-12 * abs(Ko).max()
            assert np.max(np.abs(runs[0][1] - bo)) <= 1e-13 * np.max(np.abs(bo))
