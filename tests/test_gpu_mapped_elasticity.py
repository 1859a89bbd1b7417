"""Linear elasticity on MAPPED patches (forms.ElasticityForm(geometry=...): inner(sigma(u), sym(spline.grad(v)))*spline.dx
with the Cartesian derivatives of tIGAr/common.py:1022-1040, calculusUtils.py:255-276): every block from the element
kernels of csrc/tg_assemble.hip against the oracle's element loop with explicit physical gradients, the sum-factorised
kernels against the plain one, the identity geometry against the Kronecker form, rigid-body modes, and M^T A M with the
operator implicit and the patch streamed against the oracle's product."""
import numpy as np
import pytest

from oracle import tigar_oracle as O
from geom_util import quarter_annulus, rational_volume

pytestmark = pytest.mark.gpu

LAM, MU = 1.3, 0.7


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


def _volume(T, p, nels, nfields=3):
    kvs, C = rational_volume(p, nels)
    return T.t.EqualOrderSpline(T.c.selfcomm, nfields, T.N.NURBSControlMesh([p] * 3, kvs, C)), kvs


def _blocks_close(T, uks, p, dcp, cp, tol=1e-12):
    d = len(uks)
    Ao = O.mapped_elasticity_fe_system(uks, p, cp, LAM, MU)
    N = Ao.shape[0] // d
    scale = abs(Ao).max()
    out = {}
    for i in range(d):
        for j in range(d):
            B = T.dev.assemble_mapped_elasticity_block(uks, p, dcp, i, j, LAM, MU).to_scipy()
            Bo = Ao[i * N:(i + 1) * N, j * N:(j + 1) * N]
            assert B.shape == Bo.shape and B.nnz >= Bo.nnz
            assert abs(B - Bo).max() <= tol * scale, (i, j, abs(B - Bo).max() / scale)
            out[i, j] = B
    return out, Ao


@pytest.mark.parametrize("p,nels", [(1, (3, 2, 4)), (2, (3, 4, 3)), (3, (2, 3, 2)), (3, (1, 1, 1)), (2, (5, 1, 2)), (4, (2, 1, 2))])
def test_blocks_on_a_rational_volume_match_the_oracle(T, p, nels, monkeypatch):
    """3-D: p <= 3 take the sum-factorised kernels (the walk at p = 2, one element per wave at p = 1 and 3), p = 4 the
    plain kernel; the two agree, transposed blocks are transposes, and the same bits come out of a second run"""
    gen, kvs = _volume(T, p, nels, 1)
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    dcp = [f.vector() for f in gen.cpFuncs]
    blk, Ao = _blocks_close(T, uks, p, dcp, cp)
    scale = abs(Ao).max()
    for (i, j) in ((0, 1), (1, 2), (0, 2)):
        assert abs(blk[i, j] - blk[j, i].T).max() <= 1e-12 * scale
    again = T.dev.assemble_mapped_elasticity_block(uks, p, dcp, 0, 2, LAM, MU).to_scipy()
    assert np.array_equal(again.data.view(np.int64), blk[0, 2].data.view(np.int64))
    if p <= 3:
        monkeypatch.setenv("TIGAR_ASM_LEGACY", "1")
        for (i, j) in ((0, 0), (1, 2), (2, 0)):
            Bl = T.dev.assemble_mapped_elasticity_block(uks, p, dcp, i, j, LAM, MU).to_scipy()
            assert np.array_equal(Bl.indptr, blk[i, j].indptr) and np.array_equal(Bl.indices, blk[i, j].indices)
            assert abs(Bl - blk[i, j]).max() <= 1e-12 * scale
        monkeypatch.delenv("TIGAR_ASM_LEGACY")
        if p >= 2:      # the other of the two sum-factorised kernels
            monkeypatch.setenv("TIGAR_ASM_WALK", "1" if p == 3 else "0")
            for (i, j) in ((1, 1), (0, 2)):
                Bw = T.dev.assemble_mapped_elasticity_block(uks, p, dcp, i, j, LAM, MU).to_scipy()
                assert abs(Bw - blk[i, j]).max() <= 1e-12 * scale


def test_blocks_in_two_dimensions_match_the_oracle(T):
    # exact NURBS quarter annulus (p = 2) and a polynomial map at p = 3, 5 on a stretched grid
    kv, Pf = quarter_annulus(4)
    gen = T.t.EqualOrderSpline(1, T.N.NURBSControlMesh([2, 2], [kv, kv], Pf))
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(2)]
    _blocks_close(T, uks, 2, [f.vector() for f in gen.cpFuncs], [f.vector().get_local() for f in gen.cpFuncs])
    for p, nel in ((3, (3, 2)), (5, (2, 2))):
        kvs = [T.B.uniformKnots(p, 0., 1., nel[0]), T.B.uniformKnots(p, 0., 2., nel[1])]
        gen = T.t.EqualOrderSpline(1, T.B.ExplicitBSplineControlMesh([p, p], kvs))
        g = gen.V.grids[0]
        x, y = gen.cpFuncs[0].vector().get_local(), gen.cpFuncs[1].vector().get_local()
        w = 1.0 + 0.3 * x * y
        cp = [(x + 0.2 * y * y) * w, (y - 0.1 * x * y) * w, w]
        uks = [np.asarray(g.vertices[k]) for k in range(2)]
        _blocks_close(T, uks, p, [T.dev.DeviceVector(data=c) for c in cp], cp)
    # a surface in 3-D has no elasticity form here (nsd != d): refused, not computed
    with pytest.raises(T.dev.TigarHipError):
        T.dev.assemble_mapped_elasticity_block(uks, p, [T.dev.DeviceVector(data=c) for c in cp + [w]], 0, 0, LAM, MU)


@pytest.mark.parametrize("d,p,nel", [(2, 2, 4), (2, 4, 2), (3, 2, 3), (3, 3, 2)])
def test_identity_geometry_equals_the_kronecker_form(T, d, p, nel):
    kv = [T.B.uniformKnots(p, 0., 1. + 0.5 * k, nel + k) for k in range(d)]
    gen = T.t.EqualOrderSpline(d, T.B.ExplicitBSplineControlMesh([p] * d, kv))
    Ak = T.F.ElasticityForm(LAM, MU).assemble_matrix(gen.V).to_scipy()
    fm = T.F.ElasticityForm(LAM, MU, geometry=gen)
    assert fm.block_factors(gen.V) is None and fm.symmetric is True
    Am = fm.assemble_matrix(gen.V).to_scipy()
    assert np.array_equal(Ak.indptr, Am.indptr) and np.array_equal(Ak.indices, Am.indices)
    assert abs(Ak - Am).max() <= 2e-12 * abs(Ak).max()
    with pytest.raises(NotImplementedError):
        fm.assemble_matrix(gen.V, 0, 5)


def test_rigid_body_modes_of_the_physical_configuration(T):
    """polynomial map of degree <= p with weight 1: the Lagrange space holds F exactly, so rotations about the physical
    axes are in the space and carry no strain"""
    p, nels = 3, (3, 2, 3)
    kvs = [T.B.uniformKnots(p, 0., 1., n) for n in nels]
    gen0 = T.t.EqualOrderSpline(1, T.B.ExplicitBSplineControlMesh([p] * 3, kvs))
    g = gen0.V.grids[0]
    X = [gen0.cpFuncs[i].vector().get_local() for i in range(3)]
    Y = [X[0] + 0.2 * X[1] * X[2], X[1] + 0.3 * X[0] ** 2, X[2] * (1 + 0.25 * X[0]) - 0.1 * X[1] ** 3]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    dcp = [T.dev.DeviceVector(data=c) for c in Y + [np.ones_like(X[0])]]
    N = X[0].size
    blocks = [[T.dev.assemble_mapped_elasticity_block(uks, p, dcp, i, j, LAM, MU) for j in range(3)] for i in range(3)]
    A = T.dev.csr_from_blocks(blocks)
    As = A.to_scipy()
    scale = abs(As).max()
    assert abs(As - As.T).max() <= 1e-12 * scale
    modes = [np.concatenate([np.ones(N) if f == k else np.zeros(N) for f in range(3)]) for k in range(3)]
    for (i, j) in ((0, 1), (1, 2), (0, 2)):
        u = np.zeros(3 * N)
        u[i * N:(i + 1) * N] = -Y[j]
        u[j * N:(j + 1) * N] = Y[i]
        modes.append(u)
    for u in modes:
        assert np.max(np.abs(A.mult(T.dev.DeviceVector(data=u)).get_local())) <= 1e-11 * scale


def _clamp(gen, nf):
    sp0 = gen.getScalarSpline(0)
    for f in range(nf):
        gen.addZeroDofs(f, sp0.getSideDofs(0, 0))


@pytest.mark.parametrize("p,nels,sub,implicit", [(2, (3, 3, 5), 2, True), (3, (2, 3, 4), 3, True), (2, (3, 2, 3), 0, False)])
def test_mapped_elasticity_through_the_spline(T, p, nels, sub, implicit, monkeypatch):
    """assembleMatrix of the mapped form on three fields: with M assembled (one device, whole blocks) and with the operator
    implicit and the patch streamed in sub-slabs (the form hands out row blocks of its field blocks on control-function
    windows) -- against the oracle's M^T A M with the block-diagonal M; then the solve against a direct solve"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    if implicit:
        monkeypatch.setenv("TIGAR_IMPLICIT_M", "1")
        monkeypatch.setenv("TIGAR_SUB_PLANES", str(sub))
    gen, kvs = _volume(T, p, nels)
    _clamp(gen, 3)
    assert bool(getattr(gen.M, "is_implicit", False)) == implicit
    spline = T.t.ExtractedSpline(gen, 2 * p, comm=gen.comm)
    form = T.F.ElasticityForm(LAM, MU, geometry=gen)
    T.dev.prof_reset()
    K = spline.assembleMatrix(form, diag=1.5)
    walks = T.dev.prof_get(5)[1]
    Ks = K.to_scipy()
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    Ao = O.mapped_elasticity_fe_system(uks, p, cp, LAM, MU)
    s = O.BSpline([p] * 3, [list(k) for k in kvs])
    Mo = O.generate_M_tensor(s, nfields=3)
    zd = [int(i) for i in gen.zeroDofsArray()] if hasattr(gen, "zeroDofsArray") else list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, Ao, zd, diag=1.5)
    if implicit:         # several fields streamed: K in the numbering of the slab engine (dof planes, fields inside)
        n2o = spline._slab_path().new_of_old()
        old_of_new = np.empty(Ko.shape[0], dtype=np.int64)
        old_of_new[n2o] = np.arange(Ko.shape[0])
        Ko = Ko[spline.localDofIndices()][:, old_of_new].tocsr()
    assert Ks.shape == Ko.shape
    assert abs(Ks - Ko).max() <= 1e-12 * abs(Ko).max()
    assert abs(Ks - Ks.T).max() <= 1e-12 * abs(Ko).max()
    if implicit:
        assert walks > 0                                   # the tensor line walks took the blocks on their certificate
    # gravity along -z on the physical configuration: b_z = -Mass 1
    N = cp[0].size
    Mm, _, _ = O.mapped_fe_system(uks, p, cp)
    b = np.concatenate([np.zeros(2 * N), -(Mm @ np.ones(N))])
    rhs_o = O.extract_vector(Mo, b, zd)
    Uo = spla.spsolve(sp.csc_matrix(Ko), rhs_o)
    if not implicit:
        rhs = spline.extractVector(b)
        solver = T.t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-12
        spline.setSolverOptions(linearSolver=solver)
        u = T.t.Function(spline.V)
        U = spline.solveLinearSystem(K, rhs, u)
        assert np.max(np.abs(U.get_local() - Uo)) <= 1e-8 * np.max(np.abs(Uo))
