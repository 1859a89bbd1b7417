"""Three displacement fields on one 3-D tensor basis (EqualOrderSpline(3, ...), tIGAr/common.py:1891-1914): the FE matrix
of linear elasticity assembled block by block by the Kronecker-sum kernel (forms.ElasticityForm) against an element-loop
oracle, M^T A M through the public API (scalar tensor-pattern passes per field block) against the oracle's product with the
block-diagonal M, and the solved problem against a direct solve of the oracle's system."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,nels", [(2, (3, 2, 4)), (3, (2, 3, 2)), (1, (4, 5, 3))])
def test_elasticity_three_fields(p, nels):
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev
    lam, mu = 1.3, 0.7
    ends = [(0., 1.), (0., 2.), (-1., 1.)]
    kvs = [B.uniformKnots(p, a, b, n) for (a, b), n in zip(ends, nels)]
    gen = t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh([p] * 3, kvs))
    sp0 = gen.getScalarSpline(0)
    for f in range(3):
        gen.addZeroDofs(f, sp0.getSideDofs(0, 0))                    # clamped at x = 0
    spline = t.ExtractedSpline(gen, 2 * p)
    form = F.ElasticityForm(lam, mu)
    A = form.assemble_matrix(spline.V)
    uks = [np.linspace(a, b, n + 1) for (a, b), n in zip(ends, nels)]
    Ao = O.elasticity_fe_system(uks, p, lam, mu)
    Ad = A.to_scipy()
    assert Ad.shape == Ao.shape and Ad.nnz == 9 * np.prod([O.fe_1d_matrices(u, p)[0].nnz for u in uks])
    assert abs(Ad - Ao).max() <= 1e-13 * abs(Ao).max()
    # rigid-body modes lie in the null space of the unconstrained operator
    grid = spline.V.grids[0]
    X = grid.coordinates()
    N = X.shape[0]
    for mode in (np.concatenate([np.ones(N), np.zeros(2 * N)]), np.concatenate([-X[:, 1], X[:, 0], np.zeros(N)])):
        assert np.max(np.abs(A.mult(dev.DeviceVector(data=mode)).get_local())) <= 1e-12 * abs(Ao).max()
    # extraction: block-by-block tensor passes against the oracle's M^T A M with the block-diagonal M
    s = O.BSpline([p] * 3, [O.uniform_knots(p, a, b, n) for (a, b), n in zip(ends, nels)])
    Mo = O.generate_M_tensor(s, nfields=3)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, Ao, zd)
    K = spline.extractMatrix(A)
    Ks = K.to_scipy()
    # pattern: the structural one of the product (what PETSc's symbolic MatPtAP allocates [ext]): nine copies of the scalar
    # band pattern.  scipy's product, which the oracle uses, does not store results that come out as exactly 0.0 -- the
    # mixed-derivative blocks have such entries (antisymmetric 1-D factor) -- so its pattern is a subset of ours.
    k1 = [(n + p) * (2 * p + 1) - p * (p + 1) for n in nels]
    assert Ks.nnz == 9 * int(np.prod(k1))
    diff = (Ks - Ko).tocsr()
    assert abs(diff).max() <= 1e-12 * np.max(np.abs(Ko.data))
    Pk = sp.csr_matrix((np.ones(Ks.nnz), Ks.indices, Ks.indptr), shape=Ks.shape)
    Po = sp.csr_matrix((np.ones(Ko.nnz), Ko.indices, Ko.indptr), shape=Ko.shape)
    outside = (Po - Po.multiply(Pk)).tocsr()
    outside.eliminate_zeros()
    assert outside.nnz == 0 and Ko.nnz <= Ks.nnz
    # body force (0, 0, -1): nodal load b = blockdiag(Mass) f, solve, compare with a direct solve of the oracle system
    mass = O.kron_dir0_fastest([O.fe_1d_matrices(u, p)[0] for u in uks])
    b = np.concatenate([np.zeros(2 * N), -(mass @ np.ones(N))])
    rhs = spline.extractVector(b)
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-12
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    rhs_o = O.extract_vector(Mo, b, zd)
    Uo = spla.spsolve(sp.csc_matrix(Ko), rhs_o)
    assert np.max(np.abs(U.get_local() - Uo)) <= 1e-8 * np.max(np.abs(Uo))
    uz = u.vector().get_local()[2 * N:]
    assert uz.min() < 0.0 and abs(uz[X[:, 0] == 0.0]).max() < 1e-12          # sags under its weight, fixed at the clamp
