"""GPU parity tests for the remaining BASELINE configs at sizes the oracle finishes in seconds:
cfg4 (2-D p=4 C^1 biharmonic, two clamped layers, demos/biharmonic/biharmonic.py) and cfg5
(NURBS geometry through M_control*w, three unknown fields, GMRES), plus the DG node grid."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import tigar_oracle as O

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


def _biharmonic(T, nel, p=4):
    """demos/biharmonic/biharmonic.py:46-66,100-122: (-1,1)^2, u = v = 0 and du/dn = 0 through two
    layers of zero dofs, manufactured solution (cos(pi x)+1)(cos(pi y)+1)."""
    B, t, F = T.B, T.t, T.F
    kv = [B.uniformKnots(p, -1.0, 1.0, nel) for _ in range(2)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p, p], kv))
    sp0 = gen.getScalarSpline(0)
    for direction in (0, 1):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side, nLayers=2))
    spline = t.ExtractedSpline(gen, 2 * p)
    c = lambda x: np.cos(np.pi * x)
    c1 = lambda x: np.cos(np.pi * x) + 1.0
    pi4 = np.pi ** 4
    # lap^2 u = pi^4 cos(pi x)(cos(pi y)+1) + 2 pi^4 cos(pi x)cos(pi y) + pi^4 (cos(pi x)+1)cos(pi y)
    load = F.SumOfSeparableLoads([([c, c1], pi4), ([c, c], 2 * pi4), ([c1, c], pi4)])
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-12
    solver.parameters["maximum_iterations"] = 20000
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    U = spline.solveLinearVariationalProblem(F.Equation(F.BiharmonicForm(), load), u)
    return gen, spline, U, u, solver


def test_cfg4_biharmonic_matches_oracle_and_converges(T):
    p = 4
    errs = []
    for nel in (4, 8):
        gen, spline, U, u, solver = _biharmonic(T, nel, p)
        s = O.BSpline([p, p], [O.uniform_knots(p, -1., 1., nel)] * 2)
        Mo = O.generate_M_tensor(s)
        M = gen.M.to_scipy()
        assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices)
        assert np.array_equal(M.data, Mo.data)                      # p=4 extraction bit-exact
        Ao = O.biharmonic_fe_system_2d(s)
        A = T.F.BiharmonicForm().assemble_matrix(spline.V).to_scipy()
        assert abs(A - Ao).max() <= 1e-11 * abs(Ao).max()
        zd = list(spline.zeroDofs)
        zo = []
        for direction in (0, 1):
            for side in (0, 1):
                zo += s.getSideDofs(direction, side, 2)
        assert zd == zo                                              # duplicates (corners) kept
        K = spline.extractMatrix(A).to_scipy()
        Ko = O.extract_matrix(Mo, Ao, zo)
        assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
        assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
        # solution vs direct solve of the oracle system with the same load
        X, _ = O.fe_node_grid(s)
        bo = None
        uk = s.splines[0].uniqueKnots
        c = lambda x: np.cos(np.pi * x)
        c1 = lambda x: np.cos(np.pi * x) + 1.0
        pi4 = np.pi ** 4
        for (fx, fy, sc) in ((c, c1, pi4), (c, c, 2 * pi4), (c1, c, pi4)):
            term = sc * np.kron(O.fe_1d_load(uk, p, fy), O.fe_1d_load(uk, p, fx))
            bo = term if bo is None else bo + term
        Uo, uo = O.solve_linear_system(Mo, Ko, O.extract_vector(Mo, bo, zo), "direct")
        assert np.linalg.norm(U.get_local() - Uo) <= 1e-7 * np.linalg.norm(Uo)
        exact = (np.cos(np.pi * X[:, 0]) + 1.0) * (np.cos(np.pi * X[:, 1]) + 1.0)
        errs.append(np.max(np.abs(u.vector().get_local() - exact)))
    assert errs[1] < errs[0] / 8.0            # high-order convergence under refinement
    assert errs[1] < 1e-3


def _refine_control_net(coarse_spline, fine_spline, Pw):
    """Homogeneous control net of the same geometry on a refined knot vector: solve the
    interpolation problem at the fine Greville points (exact, the spaces are nested)."""
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


def _elevate(Pw):
    """degree elevation by one of a Bezier curve given by homogeneous control points [n+1, c]"""
    n = Pw.shape[0] - 1
    out = np.zeros((n + 2, Pw.shape[1]))
    out[0], out[-1] = Pw[0], Pw[-1]
    for i in range(1, n + 1):
        a = i / float(n + 1)
        out[i] = a * Pw[i - 1] + (1.0 - a) * Pw[i]
    return out


@pytest.mark.parametrize("p,nel", [(2, 6), (3, 128)])
def test_cfg5_nurbs_geometry_three_fields_gmres(T, p, nel):
    """Quarter annulus (exact circular arcs need rational weights): the weights enter only
    through cpFuncs = M_control * (homogeneous control net); three unknown fields; non-symmetric
    diagonally dominant FE matrix on the 3-field pattern solved with Jacobi-GMRES.
    (3, 128) is BASELINE cfg5 at its full size: 128 x 128 elements, p=3, 3 fields, 444 675 FE rows,
    51 483 DoFs, nnz(A) = 33 212 169 over the 9 field blocks."""
    t, N, dev = T.t, T.N, T.dev
    # coarse exact geometry: radial (linear) x angular (quadratic rational arc), degree-elevated to p
    w = 1.0 / np.sqrt(2.0)
    arc = np.array([[1.0, 0.0, 1.0], [w, w, w], [0.0, 1.0, 1.0]])         # (w x, w y, w), radius 1
    rad = np.array([[1.0], [2.0]])                                          # radii 1..2, linear
    while arc.shape[0] < p + 1:
        arc = _elevate(arc)
    while rad.shape[0] < p + 1:
        rad = _elevate(rad)
    rad = rad[:, 0]
    coarse = [O.BSpline1(p, [0] * (p + 1) + [1] * (p + 1)) for _ in range(2)]
    fine_kv = O.uniform_knots(p, 0., 1., nel)
    fine = [O.BSpline1(p, fine_kv) for _ in range(2)]
    # control net [i (radial), j (angular), (wx, wy, w)]
    Pw = np.zeros((p + 1, p + 1, 3))
    for i in range(p + 1):
        Pw[i, :, 0] = rad[i] * arc[:, 0]
        Pw[i, :, 1] = rad[i] * arc[:, 1]
        Pw[i, :, 2] = arc[:, 2]
    # refine direction by direction
    Pr = np.stack([_refine_control_net(coarse[0], fine[0], Pw[:, j, :]) for j in range(p + 1)], axis=1)
    Pf = np.stack([_refine_control_net(coarse[1], fine[1], Pr[i, :, :]) for i in range(Pr.shape[0])], axis=0)
    cm = N.NURBSControlMesh([p, p], [fine_kv, fine_kv], Pf)
    gen = t.EqualOrderSpline(3, cm)
    nsd = gen.getNsd()
    assert nsd == 2 and gen.getNFields() == 3
    # geometry at the FE nodes: F = cpFuncs[i] / cpFuncs[nsd] lies on circles of radius 1 + xi
    s = O.BSpline([p, p], [fine_kv, fine_kv])
    X, _ = O.fe_node_grid(s)
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    x, y, wt = cp[0] / cp[2], cp[1] / cp[2], cp[2]
    assert np.max(np.abs(np.hypot(x, y) - (1.0 + X[:, 0]))) < 1e-12
    assert wt.min() > 0.7 and wt.max() <= 1.0 + 1e-14
    if nel == 128:
        assert gen.M.shape == (444675, 51483) and gen.M.nnz == 5938947
    Mc = O.generate_M_tensor(s)
    bnet = np.stack([Pf[..., c].ravel(order="F") for c in range(3)], axis=1)
    for c in range(3):
        assert np.max(np.abs(cp[c] - Mc @ bnet[:, c])) < 1e-13 * max(1.0, np.max(np.abs(bnet[:, c])))
    # three-field extraction matrix = block diagonal of the scalar one
    Mo = O.generate_M_tensor(s, nfields=3)
    M = gen.M.to_scipy()
    assert M.shape == Mo.shape and abs(M - Mo).max() == 0
    # clamp two layers on one edge, all three fields (shell-like BC)
    for field in range(3):
        gen.addZeroDofs(field, gen.getScalarSpline(field).getSideDofs(0, 0, nLayers=2))
    spline = t.ExtractedSpline(gen, 2 * p)
    # deterministic non-symmetric, diagonally dominant FE matrix coupling the three fields
    A1, _, _, _ = O.poisson_fe_system(s)
    n1 = A1.shape[0]
    pat = (abs(A1) > 0).astype(np.float64).tocsr()
    blocks = [[None] * 3 for _ in range(3)]
    rng = np.random.default_rng(0)
    for a in range(3):
        for b in range(3):
            Bk = pat.copy()
            Bk.data = 0.05 * rng.standard_normal(Bk.nnz)
            blocks[a][b] = Bk
        blocks[a][a] = blocks[a][a] + sp.identity(n1) * 4.0
    A = sp.bmat(blocks, format="csr")
    if nel == 128:
        assert A.nnz == 33212169
    bvec = rng.standard_normal(A.shape[0])
    solver = t.PETScKrylovSolver("gmres", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-11
    spline.setSolverOptions(linearSolver=solver)
    K = spline.extractMatrix(A)
    rhs = spline.extractVector(bvec)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd)
    Ks = K.to_scipy()
    assert np.array_equal(Ks.indptr, Ko.indptr) and np.array_equal(Ks.indices, Ko.indices)
    if nel == 128:
        assert Ks.nnz == 7371225
    assert abs(Ks - Ko).max() <= 1e-12 * abs(Ko).max()
    Uo = spla.spsolve(Ko.tocsc(), O.extract_vector(Mo, bvec, zd))
    assert np.linalg.norm(U.get_local() - Uo) <= 1e-8 * np.linalg.norm(Uo)
    assert np.linalg.norm(u.vector().get_local() - Mo @ Uo) <= 1e-8 * np.linalg.norm(Mo @ Uo)
    assert solver.last["status"] == 0 and solver.last["iterations"] > 1
    # the oracle's GMRES restatement takes a comparable number of iterations
    _, ito, _ = O.gmres_jacobi(Ko, O.extract_vector(Mo, bvec, zd), rtol=1e-11)
    assert abs(solver.last["iterations"] - ito) <= max(3, ito // 5)


def test_dg_node_grid_for_discontinuous_basis(T):
    """A basis with an interior knot of multiplicity p+1 needs DG extraction
    (tIGAr/BSplines.py:419-427, tIGAr/common.py:167-185): nel*(p+1) nodes per direction."""
    B, t = T.B, T.t
    p = 2
    kv = [0, 0, 0, 0.5, 0.5, 0.5, 1, 1, 1]
    cm = B.ExplicitBSplineControlMesh([p, p], [kv, kv])
    gen = t.EqualOrderSpline(1, cm)
    assert gen.useDG() and gen.extractionElement() == "DG"
    s = O.BSpline([p, p], [kv, kv])
    Mo = O.generate_M_tensor(s, dg=True)
    M = gen.M.to_scipy()
    assert M.shape == ((2 * 3) ** 2, s.getNcp())
    assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices)
    assert np.array_equal(M.data, Mo.data)


def test_compatible_spline_fields_extract_like_oracle(T):
    """RT-type compatible B-splines (tIGAr/compatibleSplines.py:21-101): fields of different
    degrees per direction; M is the block-diagonal of the per-field extraction operators on each
    field's own Q_(max degree) node grid, M_control that of the control mesh."""
    from tigar_amd.compatibleSplines import BSplineCompat
    import scipy.sparse as sp
    B = T.B
    kv = [B.uniformKnots(2, 0., 1., 4), B.uniformKnots(2, 0., 2., 3)]
    cm = B.ExplicitBSplineControlMesh([2, 2], kv)
    for kind, degs in (("RT", [1, 1]), ("N", [1, 2])):
        gen = BSplineCompat(cm, kind, degs)
        assert gen.getNFields() == 2
        blocks = []
        for i in range(2):
            f = gen.getFieldSpline(i)
            so = O.BSpline([s1.p for s1 in f.splines], [np.asarray(s1.knots) for s1 in f.splines])
            blocks.append(O.generate_M_tensor(so))
            assert gen.getDegree(i) == so.getDegree() and gen.getNcp(i) == so.getNcp()
        Mo = sp.block_diag(blocks, format="csr")
        M = gen.M.to_scipy()
        assert M.shape == Mo.shape
        assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices)
        assert np.array_equal(M.data, Mo.data)
        Mc = gen.M_control.to_scipy()
        Mco = O.generate_M_tensor(O.BSpline([2, 2], [O.uniform_knots(2, 0., 1., 4), O.uniform_knots(2, 0., 2., 3)]))
        assert np.array_equal(Mc.data, Mco.data) and np.array_equal(Mc.indices, Mco.indices)
        # the extraction path works on the mixed space: M^T A M of a block matrix, BCs on field 0
        gen.addZeroDofs(0, gen.getFieldSpline(0).getSideDofs(0, 0))
        spline = T.t.ExtractedSpline(gen, 4)
        rng = np.random.default_rng(1)
        A = sp.random(M.shape[0], M.shape[0], density=0.02, random_state=4, format="csr") + sp.identity(M.shape[0]) * 3.0
        K = spline.extractMatrix(A.tocsr()).to_scipy()
        Ko = O.extract_matrix(Mo, A.tocsr(), list(spline.zeroDofs))
        assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()


def test_multipatch_bspline_extraction_is_patchwise(T):
    """MultiBSpline (tIGAr/BSplines.py:651-908): patches side by side (offset 2 in x), element-local
    FE nodes, DoFs numbered patch after patch; the kernel path (patch-wise tensor extraction,
    stacked) equals both the oracle's patch blocks and the generic getNodesAndEvals row loop."""
    import scipy.sparse as sp
    B, t = T.B, T.t
    patches = [B.BSpline([2, 2], [B.uniformKnots(2, 0., 3., 3), B.uniformKnots(2, 0., 1., 2)]),
               B.BSpline([2, 3], [B.uniformKnots(2, -1., 1., 2), B.uniformKnots(3, 0., 2., 3)])]
    mb = B.MultiBSpline(patches)
    assert mb.getNcp() == 5 * 4 + 4 * 6 and mb.doffsets == [0, 20] and mb.nel == 6 + 6
    assert all(abs(s1.knots[0]) == 0.0 and s1.knots[-1] == 1.0 for pt in patches for s1 in pt.splines)
    # point evaluation follows the reference's patch lookup and dof offsets
    ne = mb.getNodesAndEvals(np.array([2.25, 0.5]))
    loc = patches[1].getNodesAndEvals(np.array([0.25, 0.5]))
    assert [c for c, _ in ne] == [c + 20 for c, _ in loc] and [v for _, v in ne] == [v for _, v in loc]

    class CM(t.AbstractControlMesh):
        def getScalarSpline(self):
            return mb

        def getNsd(self):
            return 2

        def getHomogeneousCoordinate(self, node, direction):
            if direction == 2:
                return 1.0
            patch = 0 if node < 20 else 1
            s = patches[patch]
            local = node - mb.doffsets[patch]
            M = s.splines[0].getNcp()
            idx = (local % M, local // M)
            return s.splines[direction].greville(idx[direction]) + (2.0 * patch if direction == 0 else 0.0)
    gen = t.EqualOrderSpline(1, CM())
    M = gen.M.to_scipy()
    blocks = []
    for pt in patches:
        so = O.BSpline([s1.p for s1 in pt.splines], [np.asarray(s1.knots) for s1 in pt.splines])
        blocks.append(O.generate_M_tensor(so, degree=3, dg=True))
    Mo = sp.block_diag(blocks, format="csr")
    assert M.shape == Mo.shape == (gen.V.dim(), 44)
    assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices) and np.array_equal(M.data, Mo.data)
    # generic plug-in path (host row loop through getNodesAndEvals + tg_csr_from_triplets)
    X = gen.V.grids[0].coordinates()
    rows, cols, vals = [], [], []
    for I in range(X.shape[0]):
        for c, v in mb.getNodesAndEvals(X[I]):
            rows.append(I), cols.append(c), vals.append(v)
    Mg = T.dev.csr_from_triplets(X.shape[0], 44, rows, cols, vals, 1e-15).to_scipy()
    # (x + 2*patch) - 2*patch is not x in floating point: the row loop sees coordinates that went
    # through the patch shift, the kernel path evaluates at the patch-local nodes
    assert abs(Mg - M).max() <= 4e-15 and Mg.nnz == M.nnz
    # control functions reproduce the (shifted) identity map: partition of unity + Greville
    cp = [f.vector().get_local() for f in gen.cpFuncs]
    assert np.max(np.abs(cp[0] / cp[2] - X[:, 0])) < 1e-13 and np.max(np.abs(cp[1] / cp[2] - X[:, 1])) < 1e-13


def test_all_golden_compatible_spline_cases_extract_like_oracle(T):
    """the 29 RT / N cases of golden_compat.npz (five hand-picked, 24 seeded random: 2-D and 3-D, degrees 1-3, periodic
    directions; field definitions checked against the reference in the CPU suite): generateM of the mixed space is the block
    diagonal of the per-field operators, bit for bit; M^T A M of a random matrix on the mixed space against the oracle."""
    import json
    import os
    import scipy.sparse as sp
    from tigar_amd.compatibleSplines import BSplineCompat
    B = T.B
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_compat.npz"))
    for m in json.loads(str(g["meta"])):
        name = m["name"]
        ckv = [[float(v) for v in g["%s_ckv%d" % (name, j)]] for j in range(len(m["cdeg"]))]
        cm = B.ExplicitBSplineControlMesh(m["cdeg"], ckv)
        gen = BSplineCompat(cm, m["kind"], m["degrees"], m["periodicities"]) if m["periodicities"] is not None \
            else BSplineCompat(cm, m["kind"], m["degrees"])
        assert gen.getNFields() == m["nfields"]
        blocks = []
        for i in range(m["nfields"]):
            f = gen.getFieldSpline(i)
            so = O.BSpline([s1.p for s1 in f.splines], [np.asarray(s1.knots) for s1 in f.splines])
            blocks.append(O.generate_M_tensor(so))
        Mo = sp.block_diag(blocks, format="csr")
        M = gen.M.to_scipy()
        M.sort_indices()
        assert M.shape == Mo.shape, name
        assert np.array_equal(M.indptr, Mo.indptr) and np.array_equal(M.indices, Mo.indices), name
        assert np.array_equal(M.data, Mo.data), name
        if M.shape[0] <= 6000:
            gen.addZeroDofs(0, [0, 1])
            spline = T.t.ExtractedSpline(gen, 2 * max(max(m["degrees"]), 1) + 2)
            A = (sp.random(M.shape[0], M.shape[0], density=min(1.0, 30.0 / M.shape[0]), random_state=7, format="csr")
                 + sp.identity(M.shape[0]) * 3.0).tocsr()
            K = spline.extractMatrix(A).to_scipy()
            Ko = O.extract_matrix(Mo, A, list(spline.zeroDofs))
            assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max(), name
            # an ASSEMBLED matrix (every block on the element-coupling pattern of the common Q_P grid, random values): the
            # 2-D cases without periodic directions take the pair walks of round 6, block by block
            grids = spline.V.grids
            if len(m["cdeg"]) == 2 and m["periodicities"] is None and all(
                    gr.shape() == grids[0].shape() and not getattr(gr, "dg", False) for gr in grids):
                from tigar_amd import forms as F, device as dev
                V1 = type(spline.V)([grids[0]], spline.V.element)
                L = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
                rng = np.random.default_rng(len(name))
                nF = m["nfields"]
                blk = [[None] * nF for _ in range(nF)]
                for a in range(nF):
                    for b in range(nF):
                        Bm = L.copy()
                        Bm.data = Bm.data + 0.2 * rng.standard_normal(Bm.nnz)
                        blk[a][b] = Bm
                A2 = sp.bmat(blk, format="csr")
                dev.prof_reset()
                K2 = spline.extractMatrix(A2).to_scipy()
                assert dev.prof_get(5)[1] == nF * nF, name                 # every block through the walks
                K2o = O.extract_matrix(Mo, A2, list(spline.zeroDofs))
                assert np.array_equal(K2.indptr, K2o.indptr) and np.array_equal(K2.indices, K2o.indices), name
                assert abs(K2 - K2o).max() <= 1e-12 * abs(K2o).max(), name


def test_multipatch_evaluations_match_the_reference(T):
    """``MultiBSpline.getNodesAndEvals`` at 1 278 sample points (element corners, edges, interiors) of ten seeded random
    multi-patch configurations against the REFERENCE class's output (tests/golden/golden_multipatch.npz): the same columns in
    the same order, the values bit for bit (generateM of such spaces against the row loop: test_multipatch_bspline_extraction_is_patchwise)."""
    import json
    import os
    B, t = T.B, T.t
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_multipatch.npz"))
    npts = 0
    for m in json.loads(str(g["meta"])):
        name, npatch, degs = m["name"], m["npatch"], m["degrees"]
        patches = [B.BSpline(degs, [[float(v) for v in g["%s_p%d_kv%d_in" % (name, k, d)]] for d in range(2)]) for k in range(npatch)]
        mb = B.MultiBSpline(patches)
        pts, ptr, cols, vals = g[name + "_pts"], g[name + "_ptr"], g[name + "_cols"], g[name + "_vals"]
        for i in range(pts.shape[0]):
            ne = mb.getNodesAndEvals(pts[i])
            assert [int(e[0]) for e in ne] == cols[ptr[i]:ptr[i + 1]].tolist(), (name, i)
            assert np.array_equal(np.array([e[1] for e in ne]), vals[ptr[i]:ptr[i + 1]]), (name, i)
        npts += pts.shape[0]
    assert npts == 1278
