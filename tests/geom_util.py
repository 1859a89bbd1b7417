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


def rational_volume(p, nels):
    """(knot vectors, homogeneous control net [M, N, O, 4]) of a smooth rational (non-affine, non-separable) volume map
    of degree p: control points = a smooth map of the Greville points, weights varying -- a NURBS volume in the sense of
    tIGAr/NURBS.py (a control net in homogeneous coordinates), not a classical solid."""
    kvs = [np.asarray(O.uniform_knots(p, 0., 1., n), dtype=np.float64) for n in nels]
    grev = [np.array([np.sum(kv[i + 1:i + p + 1]) / p for i in range(len(kv) - p - 1)]) for kv in kvs]
    g0, g1, g2 = np.meshgrid(*grev, indexing="ij")
    x = g0 + 0.15 * g1 * g2
    y = g1 + 0.2 * g0 ** 2 - 0.1 * g2
    z = g2 * (1.0 + 0.3 * g0) + 0.05 * np.sin(2.0 * g1)
    w = 1.0 + 0.25 * g0 * g1 + 0.1 * g2
    return kvs, np.stack([w * x, w * y, w * z, w], axis=-1)


def quarter_cylinder_shell(nel, nel_z=None):
    """(knot vectors, homogeneous control net) of the thick quarter cylinder 1 <= r <= 2, 0 <= theta <= pi/2, 0 <= z <= 1:
    the exact NURBS quarter annulus (degree 2) extruded along z with a degree-2 B-spline direction."""
    nel_z = nel if nel_z is None else nel_z
    kv, Pf = quarter_annulus(nel)
    kz = np.asarray(O.uniform_knots(2, 0., 1., nel_z), dtype=np.float64)
    gz = np.array([np.sum(kz[i + 1:i + 3]) / 2.0 for i in range(len(kz) - 3)])
    C = np.zeros(Pf.shape[:2] + (len(gz), 4))
    C[..., 0] = Pf[:, :, None, 0]
    C[..., 1] = Pf[:, :, None, 1]
    C[..., 2] = Pf[:, :, None, 2] * gz[None, None, :]
    C[..., 3] = Pf[:, :, None, 2]
    return [np.asarray(kv, dtype=np.float64), np.asarray(kv, dtype=np.float64), kz], C
