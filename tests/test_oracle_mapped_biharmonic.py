"""The element-loop oracle of the biharmonic form on a MAPPED patch (test infrastructure for
forms.BiharmonicForm(geometry=...)): on a stretched and on a rotated box it is the Kronecker oracle of the identity
geometry, lap of functions the space holds comes out exactly on a polynomial map and on the NURBS annulus (lap r^2 = 4), and
the energy error of an interpolant drops at first order (p = 2)."""
import numpy as np

from oracle import tigar_oracle as O


def _nodes(uks, p):
    nn = [(len(u) - 1) * p + 1 for u in uks]
    ax = [np.interp(np.arange(n) / p, np.arange(len(u)), u) for n, u in zip(nn, uks)]
    X = np.stack([g.ravel(order="F") for g in np.meshgrid(*ax, indexing="ij")], axis=1)
    return X, int(np.prod(nn))


def test_affine_maps_reproduce_the_identity_oracle():
    p = 3
    kv = [O.uniform_knots(p, -1., 1., 3), O.uniform_knots(p, 0., 2., 2)]
    s = O.BSpline([p, p], kv)
    Ai = O.biharmonic_fe_system_2d(s)
    uks = [s.splines[k].uniqueKnots for k in range(2)]
    X, N = _nodes(uks, p)
    A = O.mapped_biharmonic_fe_system(uks, p, [X[:, 0], X[:, 1], np.ones(N)])
    assert abs(A - Ai).max() <= 1e-11 * abs(Ai).max()
    # rotation + translation + constant weight: the Laplacian does not notice
    th = 0.6
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    Y = X @ R.T + np.array([0.3, -2.0])
    Ar = O.mapped_biharmonic_fe_system(uks, p, [1.7 * Y[:, 0], 1.7 * Y[:, 1], 1.7 * np.ones(N)])
    assert abs(Ar - Ai).max() <= 1e-11 * abs(Ai).max()
    # a stretched box: the identity oracle of the image mesh
    sc = np.array([2.0, 0.5])
    s2 = O.BSpline([p, p], [list(sc[k] * np.asarray(kv[k])) for k in range(2)])
    As = O.mapped_biharmonic_fe_system(uks, p, [sc[0] * X[:, 0], sc[1] * X[:, 1], np.ones(N)])
    assert abs(As - O.biharmonic_fe_system_2d(s2)).max() <= 1e-11 * abs(As).max()


def test_lap_of_functions_in_the_space_on_a_polynomial_map():
    """x = xi + 0.2 eta^2, y = eta - 0.1 xi (weight 1): polynomials in (x, y) whose pull-back stays in Q_3 are in the
    space, so their Laplacian comes out to rounding: 1, x, y, x y (harmonic), y^2 (lap = 2)"""
    p = 3
    uks = [np.linspace(0, 1, 3), np.linspace(0, 1, 4)]
    X, N = _nodes(uks, p)
    x = X[:, 0] + 0.2 * X[:, 1] ** 2
    y = X[:, 1] - 0.1 * X[:, 0]
    A, L, s, Xq = O.mapped_biharmonic_fe_system(uks, p, [x, y, np.ones(N)], return_operator=True)
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    for u in (np.ones(N), x, y, 2 * x - 3 * y + 1):
        assert np.max(np.abs(L @ u)) <= 1e-9 * np.max(np.abs(L.data))
        assert np.max(np.abs(A @ u)) <= 1e-9 * abs(A).max()
    # x y = (xi + 0.2 eta^2)(eta - 0.1 xi) has degrees (2, 3): in Q_3, and harmonic
    assert np.max(np.abs(L @ (x * y))) <= 1e-9 * np.max(np.abs(L.data))
    # u = x^2 has degree (2, 4) in (xi, eta): not in Q_3 -- but y^2 has (2, 2): lap(y^2) = 2
    assert np.max(np.abs(L @ (y * y) - 2.0)) <= 1e-9 * np.max(np.abs(L.data))
    # area: det DF = 1 + 0.04 eta, integrated over the unit square
    assert abs(float(s.sum()) - 1.02) < 1e-12


def test_lap_on_the_nurbs_annulus_converges():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from geom_util import quarter_annulus
    errs = []
    for nel in (2, 4, 8):
        kv, Pf = quarter_annulus(nel)
        s = O.BSpline([2, 2], [kv, kv])
        M = O.generate_M_tensor(s)
        uks = [s.splines[k].uniqueKnots for k in range(2)]
        cp = [M @ Pf[:, :, i].ravel(order="F") for i in range(3)]
        A, L, sw, Xq = O.mapped_biharmonic_fe_system(uks, 2, cp, return_operator=True)
        # r^2 = (1 + xi)^2 is in the space (the radial direction of this parametrisation is linear in xi): lap(r^2) = 4
        # to rounding, through a rational map with weights that vary in theta
        X = np.stack([cp[0] / cp[2], cp[1] / cp[2]], axis=1)
        r2 = X[:, 0] ** 2 + X[:, 1] ** 2
        assert np.max(np.abs(L @ r2 - 4.0)) <= 1e-10
        # the nodal interpolant of u = x^3 (lap = 6 x): first order in the energy norm at p = 2
        errs.append(np.sqrt(np.sum(sw * (L @ X[:, 0] ** 3 - 6.0 * Xq[:, 0]) ** 2)))
        assert abs(float(sw.sum()) - 0.75 * np.pi) < 1e-3 / nel ** 2
    assert errs[1] < errs[0] / 1.6 and errs[2] < errs[1] / 1.8
