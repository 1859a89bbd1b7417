"""The element-loop elasticity oracle (test infrastructure for forms.ElasticityForm): symmetry, rigid-body modes in the null
space, the energy of a uniaxial strain, and its diagonal blocks against the Kronecker form of the scalar Laplacian."""
import numpy as np

from oracle import tigar_oracle as O


def test_elasticity_oracle_identities():
    uks = [np.linspace(0, 1, 3), np.linspace(0, 2, 4), np.linspace(-1, 1, 3)]
    p, lam, mu = 2, 1.3, 0.7
    A = O.elasticity_fe_system(uks, p, lam, mu)
    nn = [(len(u) - 1) * p + 1 for u in uks]
    N = int(np.prod(nn))
    ax = [np.interp(np.arange(n) / p, np.arange(len(u)), u) for n, u in zip(nn, uks)]
    X = np.stack([g.ravel(order="F") for g in np.meshgrid(*ax, indexing="ij")], axis=1)
    assert A.shape == (3 * N, 3 * N) and abs(A - A.T).max() < 1e-14
    scale = abs(A).max()
    for f in range(3):
        u = np.zeros(3 * N)
        u[f * N:(f + 1) * N] = 1.0
        assert np.max(np.abs(A @ u)) < 1e-13 * scale
    for (i, j) in ((0, 1), (1, 2), (0, 2)):                       # infinitesimal rotations
        u = np.zeros(3 * N)
        u[i * N:(i + 1) * N] = -X[:, j]
        u[j * N:(j + 1) * N] = X[:, i]
        assert np.max(np.abs(A @ u)) < 1e-13 * scale
    u = np.concatenate([X[:, 0], np.zeros(2 * N)])                # eps_xx = 1: energy (lam + 2 mu) * volume
    assert abs(u @ (A @ u) - (lam + 2 * mu) * 1.0 * 2.0 * 2.0) < 1e-12
    # lam = 0, mu = 1: block (i, i) = Laplacian + int d_i d_i
    A1 = O.elasticity_fe_system(uks, p, 0.0, 1.0)
    M1 = [O.fe_1d_matrices(u, p)[0] for u in uks]
    K1 = [O.fe_1d_matrices(u, p)[1] for u in uks]
    lap = sum(O.kron_dir0_fastest([K1[k] if k == dd else M1[k] for k in range(3)]) for dd in range(3))
    for i in range(3):
        extra = O.kron_dir0_fastest([K1[k] if k == i else M1[k] for k in range(3)])
        blk = A1[i * N:(i + 1) * N, i * N:(i + 1) * N]
        assert abs(blk - (lap + extra)).max() < 1e-13 * abs(lap).max()
