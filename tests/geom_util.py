"""Geometry helpers shared by the GPU tests: exact NURBS quarter annulus (radii 1..2), knot-refined
by interpolation at the fine Greville points (exact: the spaces are nested)."""
import numpy as np

from oracle import tigar_oracle as O


def _refine_control_net(coarse_spline, fine_spline, Pw):
    nf = fine_spline.getNcp()
    g = np.array([fine_spline.greville(i) for i in range(nf)])
    Nf = np.zeros((nf, nf))
    Nc = np.zeros((nf, coarse_spline.getNcp()))
    for r, u in enumerate(g):
        sf = fine_spline.getKnotSpan(u)
        Nf[r, fine_spline.getNodes(u)] = fine_spline.basisFuncs(sf, u)
        sc = coarse_spline.getKnotSpan(u)
        Nc[r, coarse_spline.getNodes(u)] = coarse_spline.basisFuncs(sc, u)
    return np.linalg.solve(Nf, Nc @ Pw)


def quarter_annulus(nel, p=2):
    """(knots, homogeneous control net [ncp_r, ncp_theta, 3]) of the quarter annulus 1 <= r <= 2,
    0 <= theta <= pi/2; direction 0 radial, direction 1 angular; degree 2."""
    assert p == 2
    w = 1.0 / np.sqrt(2.0)
    arc = np.array([[1.0, 0.0, 1.0], [w, w, w], [0.0, 1.0, 1.0]])
    rad = np.array([1.0, 1.5, 2.0])
    coarse = [O.BSpline1(p, [0, 0, 0, 1, 1, 1]) for _ in range(2)]
    kv = O.uniform_knots(p, 0., 1., nel)
    fine = [O.BSpline1(p, kv) for _ in range(2)]
    Pw = np.zeros((3, 3, 3))
    for i in range(3):
        Pw[i, :, 0] = rad[i] * arc[:, 0]
        Pw[i, :, 1] = rad[i] * arc[:, 1]
        Pw[i, :, 2] = arc[:, 2]
    Pr = np.stack([_refine_control_net(coarse[0], fine[0], Pw[:, j, :]) for j in range(3)], axis=1)
    Pf = np.stack([_refine_control_net(coarse[1], fine[1], Pr[i, :, :]) for i in range(Pr.shape[0])], axis=0)
    return kv, Pf
