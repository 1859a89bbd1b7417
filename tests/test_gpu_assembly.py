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


# ---- round 5: sum-factorised element kernel (3-D), row blocks, the streamed and the multi-rank path ---------------------
def _volume_generator(T, p, nels, comm=None):
    from geom_util import rational_volume
    from tigar_amd import common as tc
    kvs, C = rational_volume(p, nels)
    cm = T.N.NURBSControlMesh([p] * 3, kvs, C)
    return T.t.EqualOrderSpline(comm if comm is not None else tc.selfcomm, 1, cm), kvs


@pytest.mark.parametrize("p,nels", [(1, (3, 2, 4)), (2, (3, 4, 3)), (3, (2, 3, 2)), (3, (1, 1, 1)), (2, (5, 1, 2))])
def test_sum_factorised_element_kernel_matches_oracle(T, p, nels, monkeypatch):
    """3-D patches with p + 1 Gauss points per direction take the wave-per-element kernel (k_asf3): against the oracle's
    plain element loop on a rational volume map, and against the plain O((p+1)^9) kernel it replaces"""
    gen, kvs = _volume_generator(T, p, nels)
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    Mo, Ko, _ = O.mapped_fe_system(uks, p, cp)
    dcp = [f.vector() for f in gen.cpFuncs]
    Mf = T.dev.assemble_mapped_matrix(uks, p, dcp, "mass")
    Kf = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace")
    _close(Mf, Mo)
    _close(Kf, Ko)
    monkeypatch.setenv("TIGAR_ASM_LEGACY", "1")
    Kl = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy()
    monkeypatch.delenv("TIGAR_ASM_LEGACY")
    Kf = Kf.to_scipy()
    assert np.array_equal(Kl.indptr, Kf.indptr) and np.array_equal(Kl.indices, Kf.indices)
    assert abs(Kl - Kf).max() <= 1e-12 * abs(Kl).max()
    # bit-reproducible (stored by the first contributor, added colour by colour)
    K2 = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy()
    assert np.array_equal(K2.data.view(np.int64), Kf.data.view(np.int64))


@pytest.mark.parametrize("p,nels", [(3, (3, 2, 4)), (2, (3, 3, 5)), (4, (2, 1, 3))])
def test_row_blocks_with_control_function_windows(T, p, nels):
    """rows of whole node planes from control functions given on a window of planes only: bit for bit the rows of the
    whole matrix / vector (p = 4: the plain kernel, which takes the same arguments)"""
    import scipy.sparse as sps
    gen, kvs = _volume_generator(T, p, nels)
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    n0, n1, n2 = g.shape()
    plane = n0 * n1
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    fn = np.sin(3.0 * cp[0]) + cp[1]
    dcp = [f.vector() for f in gen.cpFuncs]
    whole = {f: T.dev.assemble_mapped_matrix(uks, p, dcp, f).to_scipy() for f in ("mass", "laplace")}
    bw = T.dev.assemble_mapped_load(uks, p, dcp, T.dev.DeviceVector(data=fn)).get_local()
    cuts = [0, 1, p, p + 2, 2 * p, n2 - 1, n2]
    cuts = sorted(set(c for c in cuts if 0 <= c <= n2))
    parts = {"mass": [], "laplace": []}
    bparts = []
    for za, zb in zip(cuts[:-1], cuts[1:]):
        e0 = za // p - 1 if (za > 0 and za % p == 0) else za // p
        e1 = min(nels[2], (zb - 1) // p + 1)
        fa, fb = e0 * p, e1 * p + 1
        win = [T.dev.DeviceVector(data=c[fa * plane:fb * plane]) for c in cp]
        for f in parts:
            parts[f].append(T.dev.assemble_mapped_matrix(uks, p, win, f, row0=za * plane, row1=zb * plane,
                                                         cp_node0=fa * plane).to_scipy())
        bparts.append(T.dev.assemble_mapped_load(uks, p, win, T.dev.DeviceVector(data=fn[fa * plane:fb * plane]),
                                                 row0=za * plane, row1=zb * plane, cp_node0=fa * plane).get_local())
    for f in parts:
        S = sps.vstack(parts[f]).tocsr()
        assert np.array_equal(S.indptr, whole[f].indptr) and np.array_equal(S.indices, whole[f].indices)
        assert np.array_equal(S.data.view(np.int64), whole[f].data.view(np.int64))
    assert np.array_equal(np.concatenate(bparts).view(np.int64), bw.view(np.int64))
    # a window that does not cover the elements of the rows is refused
    with pytest.raises(T.dev.TigarHipError):
        T.dev.assemble_mapped_matrix(uks, p, [T.dev.DeviceVector(data=c[:plane]) for c in cp], "mass", row0=0,
                                     row1=plane * n2, cp_node0=0)


def _zero_all_faces(gen):
    sp0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))


@pytest.mark.parametrize("p,nels,sub", [(3, (3, 3, 5), 2), (2, (4, 3, 6), 3), (4, (2, 2, 3), 2)])
def test_mapped_forms_streamed_through_the_slab_engine(T, p, nels, sub, monkeypatch):
    """assembleMatrix / assembleVector of mapped forms with the operator kept implicit and the patch streamed in
    sub-slabs of dof planes (what cfg3's size needs): the forms hand out row blocks, the tensor-pattern passes take them
    on their certificate -- against the oracle's M^T A M and M^T b on the same rational volume"""
    monkeypatch.setenv("TIGAR_IMPLICIT_M", "1")
    monkeypatch.setenv("TIGAR_SUB_PLANES", str(sub))
    gen, kvs = _volume_generator(T, p, nels)
    _zero_all_faces(gen)
    assert getattr(gen.M, "is_implicit", False)
    spline = T.t.ExtractedSpline(gen, 2 * p, comm=gen.comm)
    T.dev.prof_reset()
    K = spline.assembleMatrix(T.F.LaplaceForm(geometry=gen), diag=2.0).to_scipy()
    walks = T.dev.prof_get(5)[1]
    certified = T.dev.prof_get(3)[1]
    b = spline.assembleVector(T.F.NodalLoadForm(lambda x: np.sin(x[:, 0]) + x[:, 1] * x[:, 2], gen)).get_local()
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    X = np.stack([cp[i] / cp[3] for i in range(3)], axis=1)
    _, Ko, bo = O.mapped_fe_system(uks, p, cp, fnodal=np.sin(X[:, 0]) + X[:, 1] * X[:, 2])
    s = O.BSpline([p] * 3, [list(k) for k in kvs])
    Mo = O.generate_M_tensor(s)
    zd = [int(i) for i in gen.zeroDofsArray()]
    Kr = O.extract_matrix(Mo, Ko, zd, diag=2.0)
    br = O.extract_vector(Mo, bo, zd)
    assert np.array_equal(K.indptr, Kr.indptr) and np.array_equal(K.indices, Kr.indices)
    assert abs(K - Kr).max() <= 1e-12 * abs(Kr).max()
    assert np.max(np.abs(b - br)) <= 1e-12 * np.max(np.abs(br))
    if p <= 3:                                   # (3-D quartics: the general stages)
        assert walks > 0 and certified > 0       # the line walks ran, on the certificate of the assembled row blocks


@pytest.mark.parametrize("nels,subs", [((80, 72, 64), (7, 12)), ((256, 256, 256), (12, 9))])
def test_mapped_forms_at_scale_keep_their_invariants(T, monkeypatch, nels, subs):
    """p = 3 on a rational volume map, streamed in sub-slabs, at 80 x 72 x 64 elements (10 M FE nodes) and at the benchmark's
    256^3 (454 M FE nodes, 17.4 M dofs) -- far beyond what the oracle's element loop does in seconds; properties that hold at
    any size: constants are in the kernel of the stiffness form (M reproduces constants: K 1 = M^T A 1_fe = 0), the mass form
    and the load vector of f = 1 integrate the same functions (K_mass 1 = M^T b), symmetry, the same bits from another
    sub-slab size, and the volume against the oracle's on a COARSE mesh of the same map family."""
    monkeypatch.setenv("TIGAR_IMPLICIT_M", "1")
    p = 3
    out = {}
    for sub in subs:
        monkeypatch.setenv("TIGAR_SUB_PLANES", str(sub))
        gen, kvs = _volume_generator(T, p, nels)
        spline = T.t.ExtractedSpline(gen, 2 * p, comm=gen.comm)
        rng = np.random.default_rng(5)
        T.dev.prof_reset()
        K = spline.assembleMatrix(T.F.LaplaceForm(geometry=gen), applyBCs=False)
        assert T.dev.prof_get(5)[1] > 0                                    # the tensor line walks ran
        n = K.shape[0]
        one = T.dev.DeviceVector(data=np.ones(n))
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        dx, dy = T.dev.DeviceVector(data=x), T.dev.DeviceVector(data=y)
        kx, ky = K.mult(dx).get_local(), K.mult(dy).get_local()
        k1 = K.mult(one).get_local()
        assert np.max(np.abs(k1)) <= 1e-11 * np.max(np.abs(kx)), (np.max(np.abs(k1)), np.max(np.abs(kx)))
        a1, a2 = float(y @ kx), float(x @ ky)
        assert abs(a1 - a2) <= 1e-10 * (abs(a1) + abs(a2))
        head, tail = K.rows_to_scipy(0, 4000).data.copy(), K.rows_to_scipy(n - 4000, n).data.copy()
        del K
        Km = spline.assembleMatrix(T.F.MassForm(geometry=gen), applyBCs=False)
        m1 = Km.mult(one).get_local()
        a1, a2 = float(y @ Km.mult(dx).get_local()), float(x @ Km.mult(dy).get_local())
        assert abs(a1 - a2) <= 1e-10 * (abs(a1) + abs(a2))
        del Km
        bl = spline.assembleVector(T.F.NodalLoadForm(1.0, gen), applyBCs=False).get_local()
        assert np.max(np.abs(m1 - bl)) <= 1e-12 * np.max(np.abs(bl))       # int N_i * 1 either way
        vol = float(np.sum(bl))
        assert np.all(bl > 0.0)
        out[sub] = (head, tail, bl, vol)
        del spline, gen, dx, dy, one
    u, v = out[subs[0]], out[subs[1]]
    for k in range(3):
        assert np.array_equal(u[k].view(np.int64), v[k].view(np.int64))
    # the volume against the oracle's on a coarse mesh of the same family: the maps differ by O(h^2) in their control nets
    gen_c, _ = _volume_generator(T, p, (6, 6, 6))
    g = gen_c.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen_c.cpFuncs]
    _, _, bo = O.mapped_fe_system(uks, p, cp, fnodal=np.ones(cp[0].size))
    assert abs(float(np.sum(bo)) - u[3]) <= 0.02 * u[3]


def test_poisson_on_a_nurbs_volume_converges_3d(T, monkeypatch):
    """demos/poisson/poisson-nurbs.py in 3-D without FEniCS: thick quarter cylinder (exact rational geometry, degree 2),
    u = (r-1)(2-r) sin(2 theta) sin(pi z), zero on the whole boundary; streamed in sub-slabs.  Max nodal error drops
    at the optimal rate (p + 1 = 3)."""
    from geom_util import quarter_cylinder_shell
    from tigar_amd import common as tc
    monkeypatch.setenv("TIGAR_IMPLICIT_M", "1")
    monkeypatch.setenv("TIGAR_SUB_PLANES", "4")

    def exact(x):
        r, th = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0])
        return (r - 1.0) * (2.0 - r) * np.sin(2.0 * th) * np.sin(np.pi * x[:, 2])

    def rhs(x):   # -(u_rr + u_r/r + u_thth/r^2 + u_zz)
        r, th = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0])
        R = (r - 1.0) * (2.0 - r)
        lap_r = -2.0 + (3.0 - 2.0 * r) / r - 4.0 * R / r ** 2
        return -(lap_r - np.pi ** 2 * R) * np.sin(2.0 * th) * np.sin(np.pi * x[:, 2])

    errs = []
    for nel in (3, 6, 12):
        kvs, C = quarter_cylinder_shell(nel)
        gen = T.t.EqualOrderSpline(tc.selfcomm, 1, T.N.NURBSControlMesh([2, 2, 2], kvs, C))
        _zero_all_faces(gen)
        spline = T.t.ExtractedSpline(gen, 4, comm=tc.selfcomm)
        solver = T.t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-12
        spline.setSolverOptions(linearSolver=solver)
        u = T.t.Function(spline.V)
        spline.solveLinearVariationalProblem(T.F.Equation(T.F.LaplaceForm(geometry=gen), T.F.NodalLoadForm(rhs, gen)), u)
        cp = [f.vector().get_local() for f in gen.cpFuncs]
        X = np.stack([cp[i] / cp[3] for i in range(3)], axis=1)
        errs.append(np.max(np.abs(u.vector().get_local() - exact(X))))
    assert errs[1] < errs[0] / 5.0 and errs[2] < errs[1] / 5.0
    assert errs[2] < 5e-4


@pytest.mark.parametrize("kind,p,nels,world,comm", [("volume", 3, (3, 2, 6), 2, "ipc"), ("volume", 2, (3, 3, 7), 3, "host"),
                                                    ("identity", 3, (2, 3, 5), 2, "ipc")])
def test_mapped_forms_on_several_ranks(tmp_path, kind, p, nels, world, comm):
    """the patch in z-slabs over 2-3 ranks (sharing the one GPU of the test box): every rank assembles the row blocks of
    its slab from the control functions on its own window of FE planes; rows of K and M^T b, the solution and the
    iteration count against the single-rank run and -- K, M^T b -- against the oracle"""
    import os
    import sys
    import scipy.sparse as sps
    from tigar_amd.launch import spawn_local
    from tigar_amd import common as tc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import gpu_rank_worker_mapped as W
    gen, spline, K, rhs, U, u, its = W.run(tc.selfcomm, kind, p, list(nels))
    Ks, rhs, U, u = K.to_scipy(), rhs.get_local(), U.get_local(), u.vector().get_local()
    # the single-rank run against the oracle
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    X = np.stack([cp[i] / cp[3] for i in range(3)], axis=1)
    _, Ko, bo = O.mapped_fe_system(uks, p, cp, fnodal=W.load(X))
    s = O.BSpline([p] * 3, [list(sp1.knots) for sp1 in gen.getScalarSpline(0).splines])
    Mo = O.generate_M_tensor(s)
    zd = [int(i) for i in gen.zeroDofsArray()]
    Kr = O.extract_matrix(Mo, Ko, zd, diag=1.5)
    assert np.array_equal(Ks.indptr, Kr.indptr) and np.array_equal(Ks.indices, Kr.indices)
    assert abs(Ks - Kr).max() <= 1e-12 * abs(Kr).max()
    assert np.max(np.abs(rhs - O.extract_vector(Mo, bo, zd))) <= 1e-12 * np.max(np.abs(rhs))
    env = {"PYTHONPATH": root + os.pathsep + os.environ.get("PYTHONPATH", ""), "TIGAR_COMM": comm, "TIGAR_DEVICE": "0"}
    rc = spawn_local(world, [os.path.join(root, "tests", "gpu_rank_worker_mapped.py"), str(tmp_path), kind, str(p),
                             ",".join(str(n) for n in nels)], env_extra=env, port=29500 + 41 * (p * 10 + world) + len(kind))
    assert rc == 0, "a rank failed"
    cover = np.zeros(Ks.shape[0], dtype=int)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        g0, g1, r0, r1 = [int(v) for v in z["g"]]
        Kl = sps.csr_matrix((z["K_data"], z["K_indices"], z["K_indptr"]), shape=(g1 - g0, Ks.shape[1]))
        assert np.array_equal(Kl.indptr, Ks[g0:g1].indptr) and np.array_equal(Kl.indices, Ks[g0:g1].indices)
        assert abs(Kl - Ks[g0:g1]).max() <= 1e-12 * abs(Ks).max()
        assert np.max(np.abs(z["rhs"] - rhs[g0:g1])) <= 1e-12 * np.max(np.abs(rhs))
        assert np.max(np.abs(z["U"] - U[g0:g1])) <= 1e-8 * np.max(np.abs(U))
        if r1 > r0:
            assert np.max(np.abs(z["u"] - u[r0:r1])) <= 1e-8 * np.max(np.abs(u))
        assert abs(int(z["its"][0]) - its) <= 3
        cover[g0:g1] += 1
    assert np.all(cover == 1)


# ---- round 6: the p = 3 stiffness / elasticity kernel with four waves on four consecutive elements (k_asf3_quad) -----------
@pytest.mark.parametrize("nels", [(37, 2, 3), (5, 3, 2), (9, 2, 2), (33, 1, 1)])
def test_quad_element_kernel_pieces_seams_and_short_groups(T, nels, monkeypatch):
    """lines of elements that do not divide into groups of four or into pieces (a seam between pieces at 32 elements, a last
    group of one, one or two elements in all), against the oracle's element loop, the element-per-wave kernel it replaces and
    itself with other piece lengths; an elasticity block; row blocks bit for bit the rows of the whole matrix"""
    import scipy.sparse as sps
    p = 3
    gen, kvs = _volume_generator(T, p, nels)
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    n0, n1, n2 = g.shape()
    plane = n0 * n1
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    dcp = [f.vector() for f in gen.cpFuncs]
    _, Ko, _ = O.mapped_fe_system(uks, p, cp)
    K = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy()
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    K2 = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy()
    assert np.array_equal(K2.data.view(np.int64), K.data.view(np.int64))            # bit-reproducible
    for env in ({"TIGAR_ASM_QUAD": "0"}, {"TIGAR_ASM_QUAD_CHUNK": "1"}, {"TIGAR_ASM_QUAD_CHUNK": "1", "TIGAR_ASM_PRE": "0"},
                {"TIGAR_ASM_QUAD_CHUNK": "2"}, {"TIGAR_ASM_QUAD_CHUNK": "3"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        Kv = T.dev.assemble_mapped_matrix(uks, p, dcp, "laplace").to_scipy()
        for k_ in env:
            monkeypatch.delenv(k_)
        assert np.array_equal(Kv.indices, K.indices)
        assert abs(Kv - K).max() <= 1e-13 * abs(K).max(), env
    # a block of the elasticity form (the same kernel, nine coefficients per point)
    Eo = O.mapped_elasticity_fe_system(uks, p, cp, 1.3, 0.7)
    Nn = Eo.shape[0] // 3
    B12 = T.dev.assemble_mapped_elasticity_block(uks, p, dcp, 1, 2, 1.3, 0.7).to_scipy()
    assert abs(B12 - Eo[Nn:2 * Nn, 2 * Nn:3 * Nn]).max() <= 1e-12 * abs(Eo).max()
    # row blocks on windows of the control functions
    cuts = sorted(set(c for c in (0, 1, p, p + 2, 2 * p, n2 - 1, n2) if 0 <= c <= n2))
    parts = []
    for za, zb in zip(cuts[:-1], cuts[1:]):
        e0 = za // p - 1 if (za > 0 and za % p == 0) else za // p
        e1 = min(nels[2], (zb - 1) // p + 1)
        fa, fb = e0 * p, e1 * p + 1
        win = [T.dev.DeviceVector(data=c[fa * plane:fb * plane]) for c in cp]
        parts.append(T.dev.assemble_mapped_matrix(uks, p, win, "laplace", row0=za * plane, row1=zb * plane,
                                                  cp_node0=fa * plane).to_scipy())
    S = sps.vstack(parts).tocsr()
    assert np.array_equal(S.indptr, K.indptr) and np.array_equal(S.indices, K.indices)
    assert np.array_equal(S.data.view(np.int64), K.data.view(np.int64))
