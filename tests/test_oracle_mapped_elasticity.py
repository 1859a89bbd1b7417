"""The element-loop oracle of linear elasticity on a MAPPED patch (test infrastructure for
forms.ElasticityForm(geometry=...)): on a stretched box it is the identity-geometry oracle of the image mesh, under a
rotation of the patch the blocks rotate with it (isotropy), rigid-body modes of the physical configuration lie in the null
space on a polynomial map the Lagrange space holds exactly, and the form is symmetric on a rational map."""
import numpy as np
import scipy.sparse as sp

from oracle import tigar_oracle as O


def _nodes(uks, p):
    nn = [(len(u) - 1) * p + 1 for u in uks]
    ax = [np.interp(np.arange(n) / p, np.arange(len(u)), u) for n, u in zip(nn, uks)]
    X = np.stack([g.ravel(order="F") for g in np.meshgrid(*ax, indexing="ij")], axis=1)
    return X, int(np.prod(nn))


def test_stretched_box_is_the_identity_oracle_of_the_image_mesh():
    p, lam, mu = 2, 1.3, 0.7
    uks = [np.linspace(0, 1, 3), np.array([0.0, 0.3, 1.0]), np.linspace(0, 1, 2)]
    X, N = _nodes(uks, p)
    sc, sh = np.array([2.0, 0.5, 3.0]), np.array([-1.0, 0.25, 4.0])
    cp = [sc[i] * X[:, i] + sh[i] for i in range(3)] + [np.ones(N)]
    Am = O.mapped_elasticity_fe_system(uks, p, cp, lam, mu)
    Ai = O.elasticity_fe_system([sc[i] * uks[i] + sh[i] for i in range(3)], p, lam, mu)
    assert abs(Am - Ai).max() <= 1e-13 * abs(Ai).max()
    # a constant weight changes nothing (homogeneous coordinates)
    Aw = O.mapped_elasticity_fe_system(uks, p, [2.5 * c for c in cp], lam, mu)
    assert abs(Aw - Ai).max() <= 1e-13 * abs(Ai).max()


def test_rotated_patch_rotates_the_blocks():
    p, lam, mu = 2, 2.0, 0.9
    uks = [np.linspace(0, 1, 3), np.linspace(0, 2, 3)]
    X, N = _nodes(uks, p)
    th = 0.37
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    Y = X @ R.T
    Am = O.mapped_elasticity_fe_system(uks, p, [Y[:, 0], Y[:, 1], np.ones(N)], lam, mu)
    Ai = O.elasticity_fe_system(uks, p, lam, mu)
    RI = sp.kron(sp.csr_matrix(R), sp.identity(N)).tocsr()
    assert abs(Am - RI @ Ai @ RI.T).max() <= 1e-13 * abs(Ai).max()


def test_rigid_body_modes_and_symmetry_on_curved_maps():
    p, lam, mu = 3, 1.3, 0.7
    uks = [np.linspace(0, 1, 3), np.linspace(0, 1, 2), np.linspace(0, 1, 3)]
    X, N = _nodes(uks, p)
    # polynomial of degree <= p per direction, weight 1: the Lagrange space holds the map exactly
    Y = np.stack([X[:, 0] + 0.2 * X[:, 1] * X[:, 2], X[:, 1] + 0.3 * X[:, 0] ** 2, X[:, 2] * (1 + 0.25 * X[:, 0]) - 0.1 * X[:, 1] ** 3],
                 axis=1)
    A = O.mapped_elasticity_fe_system(uks, p, [Y[:, 0], Y[:, 1], Y[:, 2], np.ones(N)], lam, mu)
    scale = abs(A).max()
    assert abs(A - A.T).max() <= 1e-13 * scale
    for f in range(3):
        u = np.zeros(3 * N)
        u[f * N:(f + 1) * N] = 1.0
        assert np.max(np.abs(A @ u)) <= 1e-12 * scale
    for (i, j) in ((0, 1), (1, 2), (0, 2)):
        u = np.zeros(3 * N)
        u[i * N:(i + 1) * N] = -Y[:, j]
        u[j * N:(j + 1) * N] = Y[:, i]
        assert np.max(np.abs(A @ u)) <= 1e-12 * scale
    # uniaxial strain u = (x, 0, 0): energy (lam + 2 mu) * volume, volume from the mass form of the scalar oracle
    Mm, _, _ = O.mapped_fe_system(uks, p, [Y[:, 0], Y[:, 1], Y[:, 2], np.ones(N)])
    vol = float(np.ones(N) @ (Mm @ np.ones(N)))
    u = np.concatenate([Y[:, 0], np.zeros(2 * N)])
    assert abs(u @ (A @ u) - (lam + 2 * mu) * vol) <= 1e-11 * vol
    # rational map: symmetric, translations in the null space, diagonal blocks at lam = 0, mu = 1 = Laplacian + (d_i, d_i)
    w = 1.0 + 0.2 * X[:, 0] * X[:, 1] + 0.1 * X[:, 2]
    cp = [Y[:, 0] * w, Y[:, 1] * w, Y[:, 2] * w, w]
    Ar = O.mapped_elasticity_fe_system(uks, p, cp, lam, mu)
    assert abs(Ar - Ar.T).max() <= 1e-13 * abs(Ar).max()
    for f in range(3):
        u = np.zeros(3 * N)
        u[f * N:(f + 1) * N] = 1.0
        assert np.max(np.abs(Ar @ u)) <= 1e-12 * abs(Ar).max()
    A1 = O.mapped_elasticity_fe_system(uks, p, cp, 0.0, 1.0)
    _, Ko, _ = O.mapped_fe_system(uks, p, cp)
    tr = sum(A1[i * N:(i + 1) * N, i * N:(i + 1) * N] for i in range(3))
    assert abs(tr - 4.0 * Ko).max() <= 1e-12 * abs(Ko).max()          # sum_i (Lap + d_i d_i) = 3 Lap + Lap
