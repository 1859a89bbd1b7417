"""IGA dof permutation (generatePermutation / applyPermutation, tIGAr/common.py:407-433, 1583-1665): the product's
device path (support pattern -> transpose -> majority owner per dof -> stable argsort; column relabelling of M with
re-sorted rows; renamed zero dofs) against the oracle's restatement of the reference loop, and the invariance of the
solved problem under the renumbering."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


def _generator(nfields, d, p, nel):
    import tigar_amd as t
    from tigar_amd import BSplines as B
    kv = [B.uniformKnots(p, 0., 1., nel) for _ in range(d)]
    gen = t.EqualOrderSpline(nfields, B.ExplicitBSplineControlMesh([p] * d, kv))
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    return gen, s


@pytest.mark.parametrize("nfields,d,p,nel,nparts", [(1, 2, 2, 7, 3), (2, 2, 3, 6, 4), (1, 3, 2, 4, 5), (3, 2, 1, 5, 2)])
def test_permutation_matches_the_reference_loop(nfields, d, p, nel, nparts):
    gen, s = _generator(nfields, d, p, nel)
    X, _ = O.fe_node_grid(s)
    support = O.generate_M([s] * nfields, [X] * nfields, ignore_eps=-1.0)        # every returned node, zeros kept
    assert support.nnz == nfields * X.shape[0] * (p + 1) ** d
    fe_owner = gen.feRowOwners(nparts)
    # the default FE partition: nparts contiguous runs per field, piece r of every field on rank r
    n = X.shape[0]
    assert fe_owner.shape == (nfields * n,) and fe_owner.min() == 0 and fe_owner.max() == nparts - 1
    for f in range(nfields):
        assert np.all(np.diff(fe_owner[f * n:(f + 1) * n]) >= 0)
    # the product's own support pattern is the oracle's
    S = gen.generateM(support_only=True).to_scipy()
    assert np.array_equal(S.indptr, support.indptr) and np.array_equal(S.indices, support.indices)
    perm_o = O.generate_permutation(support, fe_owner)
    perm = np.asarray(gen.generatePermutation(nparts=nparts))
    assert np.array_equal(perm, perm_o)
    assert not np.array_equal(perm, np.arange(perm.size)) or nfields == 1       # several fields: blocks interleave
    # an arbitrary (non-contiguous) ownership, as a mesh partitioner would give it
    rng = np.random.default_rng(5)
    scattered = rng.integers(0, nparts, size=fe_owner.size).astype(np.int32)
    assert np.array_equal(np.asarray(gen.generatePermutation(nparts=nparts, fe_owner=scattered)),
                          O.generate_permutation(support, scattered))
    # one part: identity
    assert np.array_equal(np.asarray(gen.generatePermutation(nparts=1)), np.arange(perm.size))

    # applyPermutation: columns of M, M^T, zero dofs
    sp0 = gen.getScalarSpline(0)
    for f in range(nfields):
        gen.addZeroDofs(f, sp0.getSideDofs(0, 0))
        gen.addZeroDofs(f, sp0.getSideDofs(1, 1))
    zd_before = list(gen.zeroDofs)
    M_before = gen.M.to_scipy()
    Mo, zd_o = O.apply_permutation(M_before, zd_before, perm_o)
    gen.applyPermutation(nparts=nparts)
    Mp = gen.M.to_scipy()
    assert np.array_equal(Mp.indptr, Mo.indptr) and np.array_equal(Mp.indices, Mo.indices)
    assert np.array_equal(Mp.data, Mo.data)
    assert abs(gen.MT.to_scipy() - Mo.T).max() == 0
    assert list(gen.zeroDofs) == list(zd_o)
    assert np.array_equal(np.asarray(gen.permutation), perm_o)


def test_solution_does_not_depend_on_the_numbering():
    """Poisson through ExtractedSpline before and after the renumbering: K and the IGA solution are permuted, the
    FE-nodal solution M U is the same."""
    import tigar_amd as t
    from tigar_amd import forms as F
    d, p, nel, nparts = 2, 2, 12, 4
    out = []
    for permute in (False, True):
        gen, s = _generator(1, d, p, nel)
        sp0 = gen.getScalarSpline(0)
        for direction in range(d):
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
        if permute:
            rng = np.random.default_rng(11)
            gen.applyPermutation(nparts=nparts, fe_owner=rng.integers(0, nparts, size=gen.V.dim()))
            assert not np.array_equal(gen.permutation, np.arange(gen.permutation.size))
        spline = t.ExtractedSpline(gen, 2 * p)
        solver = t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-12
        spline.setSolverOptions(linearSolver=solver)
        f1 = lambda x: np.sin(np.pi * x)
        K = spline.assembleMatrix(F.LaplaceForm())
        rhs = spline.assembleVector(F.SeparableLoadForm([f1] * d, scale=d * np.pi ** 2))
        u = t.Function(spline.V)
        U = spline.solveLinearSystem(K, rhs, u)
        out.append((K.to_scipy(), U.get_local(), u.vector().get_local(), gen))
    (K0, U0, u0, _), (K1, U1, u1, gen1) = out
    perm = np.asarray(gen1.permutation)
    assert abs(K1 - K0[perm][:, perm]).max() <= 1e-12 * abs(K0).max()
    assert np.max(np.abs(U1 - U0[perm])) <= 1e-9 * np.max(np.abs(U0))
    assert np.max(np.abs(u1 - u0)) <= 1e-9 * np.max(np.abs(u0))
    x = np.linspace(0, 1, p * nel + 1)
    exact = np.outer(np.sin(np.pi * x), np.sin(np.pi * x)).ravel()
    assert np.max(np.abs(u1 - exact)) < 2e-3
