"""The biharmonic form on MAPPED patches (forms.BiharmonicForm(geometry=...): inner(lap(u), lap(v))*spline.dx with
lap = spline.div(spline.grad(.)), demos/biharmonic/biharmonic.py:100-103 on a geometry that is not the identity): the element
kernel with the second derivatives of the rational map against the oracle's point-by-point matrix calculus, the identity
geometry against the Kronecker form, M^T A M streamed, and the demo's manufactured solution on a NURBS square converging
at the energy-norm rate."""
import numpy as np
import pytest

from oracle import tigar_oracle as O
from geom_util import quarter_annulus, rational_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tigar_amd
    from tigar_amd import BSplines, forms, device, NURBS, common

    class NS:
        pass
    ns = NS()
    ns.t, ns.B, ns.F, ns.dev, ns.N, ns.c = tigar_amd, BSplines, forms, device, NURBS, common
    device.device_info()
    return ns


def _close(A, Ao, tol=1e-11):
    A = A.to_scipy()
    assert A.shape == Ao.shape and A.nnz >= Ao.nnz
    assert abs(A - Ao).max() <= tol * abs(Ao).max(), abs(A - Ao).max() / abs(Ao).max()


def nurbs_square(p, nel, amp=0.12):
    """(knots, homogeneous control net [n, n, 3]) of the square (-1, 1)^2 under a smooth rational map that is not the
    identity: control points = the Greville points moved by a field that vanishes on the boundary, weights 1 on the boundary
    and varying inside -- the boundary curves stay the straight edges in their own parametrisation"""
    kv = np.asarray(O.uniform_knots(p, -1., 1., nel), dtype=np.float64)
    grev = np.array([np.sum(kv[i + 1:i + p + 1]) / p for i in range(len(kv) - p - 1)])
    g0, g1 = np.meshgrid(grev, grev, indexing="ij")
    bub = (1.0 - g0 ** 2) * (1.0 - g1 ** 2)
    x = g0 + amp * bub * np.sin(1.5 * g1 + 0.3)
    y = g1 + amp * bub * np.cos(2.0 * g0)
    w = 1.0 + 0.3 * bub
    return kv, np.stack([w * x, w * y, w], axis=-1)


def test_two_dimensional_maps_match_the_oracle(T):
    kv, Pf = quarter_annulus(3)
    gen = T.t.EqualOrderSpline(1, T.N.NURBSControlMesh([2, 2], [kv, kv], Pf))
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(2)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    _close(T.F.BiharmonicForm(geometry=gen).assemble_matrix(gen.V), O.mapped_biharmonic_fe_system(uks, 2, cp))
    for p, nel in ((3, 3), (4, 2), (6, 1)):
        kv, C = nurbs_square(p, nel)
        gen = T.t.EqualOrderSpline(1, T.N.NURBSControlMesh([p, p], [kv, kv], C))
        g = gen.V.grids[0]
        uks = [np.asarray(g.vertices[k]) for k in range(2)]
        cp = [f.vector().get_local() for f in gen.cpFuncs]
        A = T.F.BiharmonicForm(geometry=gen).assemble_matrix(gen.V)
        _close(A, O.mapped_biharmonic_fe_system(uks, p, cp))
        As = A.to_scipy()
        assert abs(As - As.T).max() <= 1e-12 * abs(As).max()
        # constants and the physical coordinates are in the kernel only as far as the space holds them: constants always
        assert np.max(np.abs(As @ np.ones(As.shape[0]))) <= 1e-9 * abs(As).max()
    # a surface in 3-D: refused (the derivative of the pseudo-inverse is not the formula used), not computed
    x, y = cp[0] / cp[2], cp[1] / cp[2]
    dcp = [T.dev.DeviceVector(data=c) for c in (x, y, x * y, np.ones_like(x))]
    with pytest.raises(T.dev.TigarHipError):
        T.dev.assemble_mapped_matrix(uks, p, dcp, "biharmonic")


@pytest.mark.parametrize("p,nel", [(3, (4, 3)), (4, (2, 3))])
def test_identity_geometry_equals_the_kronecker_form(T, p, nel):
    kv = [T.B.uniformKnots(p, -1., 1., nel[0]), T.B.uniformKnots(p, 0., 3., nel[1])]
    gen = T.t.EqualOrderSpline(1, T.B.ExplicitBSplineControlMesh([p, p], kv))
    Ak = T.F.BiharmonicForm().assemble_matrix(gen.V).to_scipy()
    Am = T.F.BiharmonicForm(geometry=gen).assemble_matrix(gen.V).to_scipy()
    assert np.array_equal(Ak.indptr, Am.indptr) and np.array_equal(Ak.indices, Am.indices)
    assert abs(Ak - Am).max() <= 1e-11 * abs(Ak).max()


@pytest.mark.parametrize("p,nels", [(2, (2, 2, 2)), (3, (1, 2, 1))])
def test_three_dimensional_rational_volume_matches_the_oracle(T, p, nels):
    kvs, C = rational_volume(p, nels)
    gen = T.t.EqualOrderSpline(T.c.selfcomm, 1, T.N.NURBSControlMesh([p] * 3, kvs, C))
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    A = T.dev.assemble_mapped_matrix(uks, p, [f.vector() for f in gen.cpFuncs], "biharmonic")
    _close(A, O.mapped_biharmonic_fe_system(uks, p, cp))
    # row blocks on a window of control-function planes: the rows of the whole matrix, bit for bit
    n0, n1, n2 = g.shape()
    plane = n0 * n1
    za, zb = p, min(n2, 2 * p)
    e0 = za // p - 1 if (za > 0 and za % p == 0) else za // p
    e1 = min(nels[2], (zb - 1) // p + 1)
    fa, fb = e0 * p, e1 * p + 1
    win = [T.dev.DeviceVector(data=c[fa * plane:fb * plane]) for c in cp]
    Ab = T.dev.assemble_mapped_matrix(uks, p, win, "biharmonic", row0=za * plane, row1=zb * plane, cp_node0=fa * plane).to_scipy()
    Aw = A.to_scipy()[za * plane:zb * plane]
    assert np.array_equal(Ab.indices, Aw.indices) and np.array_equal(Ab.data.view(np.int64), Aw.data.view(np.int64))


def test_streamed_through_the_slab_engine(T, monkeypatch):
    monkeypatch.setenv("TIGAR_IMPLICIT_M", "1")
    monkeypatch.setenv("TIGAR_SUB_PLANES", "2")
    p, nels = 2, (3, 2, 4)
    kvs, C = rational_volume(p, nels)
    gen = T.t.EqualOrderSpline(T.c.selfcomm, 1, T.N.NURBSControlMesh([p] * 3, kvs, C))
    sp0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side, nLayers=2))
    assert getattr(gen.M, "is_implicit", False)
    spline = T.t.ExtractedSpline(gen, 2 * p, comm=gen.comm)
    K = spline.assembleMatrix(T.F.BiharmonicForm(geometry=gen), diag=3.0).to_scipy()
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    Ao = O.mapped_biharmonic_fe_system(uks, p, cp)
    Mo = O.generate_M_tensor(O.BSpline([p] * 3, [list(k) for k in kvs]))
    Ko = O.extract_matrix(Mo, Ao, [int(i) for i in gen.zeroDofsArray()], diag=3.0)
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-11 * abs(Ko).max()


def test_manufactured_solution_on_a_nurbs_square_converges(T):
    """demos/biharmonic/biharmonic.py:46-139 on a square whose parametrisation is a rational map: u = v = 0 and du/dn = 0
    through two layers of zero control variables, u = (cos(pi x) + 1)(cos(pi y) + 1), f = lap lap u; the energy error
    sqrt(int lap(u_h - u)^2) (biharmonic.py:127) drops at the rate p - 1 = 2 for cubics"""
    p = 3
    errs = []
    for nel in (4, 8, 16):
        kv, C = nurbs_square(p, nel)
        gen = T.t.EqualOrderSpline(1, T.N.NURBSControlMesh([p, p], [kv, kv], C))
        sp0 = gen.getScalarSpline(0)
        for direction in (0, 1):
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side, nLayers=2))
        spline = T.t.ExtractedSpline(gen, 2 * p)
        pi = np.pi

        def f(x):
            cx, cy = np.cos(pi * x[:, 0]), np.cos(pi * x[:, 1])
            return pi ** 4 * (4.0 * cx * cy + cx + cy)
        u = T.t.Function(spline.V)
        spline.solveLinearVariationalProblem(T.F.Equation(T.F.BiharmonicForm(geometry=gen), T.F.NodalLoadForm(f, gen)), u)
        g = gen.V.grids[0]
        uks = [np.asarray(g.vertices[k]) for k in range(2)]
        cp = [fn.vector().get_local() for fn in gen.cpFuncs]
        _, L, sw, Xq = O.mapped_biharmonic_fe_system(uks, p, cp, return_operator=True)
        cx, cy = np.cos(pi * Xq[:, 0]), np.cos(pi * Xq[:, 1])
        lap_exact = -pi ** 2 * (cx * (cy + 1.0) + (cx + 1.0) * cy)
        errs.append(float(np.sqrt(np.sum(sw * (L @ u.vector().get_local() - lap_exact) ** 2))))
        assert abs(float(sw.sum()) - 4.0) < 1e-6                       # the map keeps the square (rational integrand: quadrature error)
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert rates[0] > 1.6 and rates[1] > 1.8, (errs, rates)
