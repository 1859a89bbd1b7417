"""-m gpu: the direct solver behind ``linearSolver=None`` (dolfin's solve() = sparse LU in the reference,
tIGAr/common.py:1255-1256): banded LU with partial pivoting (csrc/tg_lu.hip) against scipy's SuperLU on systems
where Jacobi-Krylov methods fail, on the biharmonic demo flow, and the hand-over to GMRES for large systems."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


def test_banded_lu_with_pivoting_vs_superlu():
    from tigar_amd import device as dev
    rng = np.random.default_rng(1)
    for (n, kl, ku) in [(1, 0, 0), (7, 2, 1), (200, 5, 9), (1500, 40, 17), (3000, 130, 130)]:
        diags = {}
        for o in range(-kl, ku + 1):
            diags[o] = rng.standard_normal(n - abs(o))
        A = sp.diags(list(diags.values()), list(diags.keys()), shape=(n, n), format="csr")
        A = A.tolil()
        for i in range(0, n if n > 1 else 0, 3):   # tiny / zero diagonal entries: no solve without row interchanges
            A[i, i] = 0.0 if i % 2 == 0 else 1e-14
        A = A.tocsr()
        xs = rng.standard_normal(n)
        b = A @ xs
        K = dev.DeviceCSR.from_scipy(A)
        bl, bu, nb = dev.lu_band_info(K)
        assert bl <= kl and bu <= ku and nb == (2 * bl + bu + 1) * n * 8
        x = dev.DeviceVector(n)
        info = dev.lu_solve(K, dev.DeviceVector(data=b), x)
        assert info == 0
        ref = spla.spsolve(A.tocsc(), b)
        err = np.linalg.norm(x.get_local() - ref) / np.linalg.norm(ref)
        res = np.linalg.norm(A @ x.get_local() - b) / np.linalg.norm(b)
        # random band matrices are badly conditioned: backward error is the criterion (as small as SuperLU's),
        # the forward error only where the condition number allows
        res_ref = np.linalg.norm(A @ ref - b) / np.linalg.norm(b)
        assert res < 1e-10 and res <= 100 * max(res_ref, 1e-16), (n, kl, ku, res, res_ref, err)
        if n <= 200:
            assert err < 1e-6 * max(1.0, np.linalg.cond(A.toarray()) * 1e-10), (n, err)


def test_blocked_factorisation_gives_the_factors_of_the_column_by_column_one(monkeypatch):
    """Panel + trailing-column kernels (NB columns per pair of launches) apply the same operations to every entry in
    the same order as one launch per column: the solutions agree bit for bit, at every panel width, with zero and
    tiny pivots (row interchanges across panel borders) and with a singular matrix (same `info`)."""
    from tigar_amd import device as dev
    rng = np.random.default_rng(7)
    # (the band widths walk through the instantiations of the register panel: 1, 2, 3 rows per thread with 256 threads,
    # 3 with 384, then the 8-column panels; other panel widths and TIGAR_LU_PANEL_REG=0 run the panel in LDS; the widest
    # band takes 4-column panels in LDS and the trailing kernel that reads its multipliers in place)
    for (n, kl, ku) in [(7, 2, 1), (97, 5, 9), (700, 40, 17), (1500, 130, 130), (64, 63, 63), (900, 300, 20),
                        (1400, 600, 600), (1500, 1000, 30), (1800, 1300, 10), (2300, 2000, 50), (2900, 2500, 20)]:
        diags = {o: rng.standard_normal(n - abs(o)) for o in range(-kl, ku + 1)}
        A = sp.diags(list(diags.values()), list(diags.keys()), shape=(n, n), format="csr").tolil()
        for i in range(0, n, 3):
            A[i, i] = 0.0 if i % 2 == 0 else 1e-14
        A = A.tocsr()
        b = dev.DeviceVector(data=A @ rng.standard_normal(n))
        K = dev.DeviceCSR.from_scipy(A)
        monkeypatch.setenv("TIGAR_LU_BLOCKED", "0")
        x0 = dev.DeviceVector(n)
        assert dev.lu_solve(K, b, x0) == 0
        for nb in ("", "lds", "2", "4", "5", "8"):
            monkeypatch.setenv("TIGAR_LU_BLOCKED", "1")
            monkeypatch.setenv("TIGAR_LU_PANEL_REG", "0" if nb == "lds" else "1")
            nb = "" if nb == "lds" else nb
            if nb:
                monkeypatch.setenv("TIGAR_LU_NB", nb)
            else:
                monkeypatch.delenv("TIGAR_LU_NB", raising=False)
            x1 = dev.DeviceVector(n)
            assert dev.lu_solve(K, b, x1) == 0
            assert np.array_equal(x0.get_local().view(np.int64), x1.get_local().view(np.int64)), (n, kl, ku, nb)
        monkeypatch.delenv("TIGAR_LU_NB", raising=False)
        monkeypatch.delenv("TIGAR_LU_PANEL_REG", raising=False)
    # singular: a zero column inside the band -> the same first zero pivot reported
    n = 50
    A = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="lil")
    A[:, 20] = 0.0
    A[20, :] = 0.0
    K = dev.DeviceCSR.from_scipy(sp.csr_matrix(A))
    infos = []
    for blocked in ("0", "1"):
        monkeypatch.setenv("TIGAR_LU_BLOCKED", blocked)
        infos.append(dev.lu_solve(K, dev.DeviceVector(data=np.ones(n)), dev.DeviceVector(n)))
    assert infos[0] == infos[1] == 21, infos


def test_saddle_point_system_where_jacobi_krylov_stalls():
    """[[A, B], [B^T, 0]]: zero diagonal block (PCJACOBI substitutes 1), indefinite -- the kind of system the
    reference's default LU handles and a Jacobi-Krylov default does not"""
    import tigar_amd as t
    from tigar_amd import device as dev
    rng = np.random.default_rng(2)
    n1, n2 = 300, 120
    A = sp.diags([rng.random(n1) + 2.0, -np.ones(n1 - 1), -np.ones(n1 - 1)], [0, 1, -1], format="csr")
    B = sp.random(n1, n2, density=0.05, random_state=3, format="csr") + sp.eye(n1, n2, format="csr")
    K = sp.bmat([[A, B], [B.T, None]], format="csr")
    xs = rng.standard_normal(n1 + n2)
    b = K @ xs
    lu = t.PETScLUSolver()
    x = dev.DeviceVector(n1 + n2)
    lu.solve(K, x, b)
    assert np.linalg.norm(x.get_local() - xs) <= 1e-8 * np.linalg.norm(xs)
    assert lu.last["info"] == 0
    # singular system: explicit error, no garbage
    S = K.tolil()
    S[5, :] = 0.0
    with pytest.raises(RuntimeError):
        lu.solve(S.tocsr(), dev.DeviceVector(n1 + n2), b)


def test_default_solver_is_direct_on_the_biharmonic_demo_flow():
    """demos/biharmonic/biharmonic.py with the reference's default (direct) solve: p = 4, clamped, 64 x 64 elements
    (4624 DoFs), and BASELINE cfg4's size 256 x 256 (67600 DoFs, half-bandwidth 1044) against the manufactured
    solution"""
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    p = 4
    errs = {}
    for nel in (16, 64, 256):
        kv = [B.uniformKnots(p, -1.0, 1.0, nel) for _ in range(2)]
        gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p, p], kv))
        sp0 = gen.getScalarSpline(0)
        for direction in (0, 1):
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side, nLayers=2))
        spline = t.ExtractedSpline(gen, 2 * p)
        assert spline.linearSolver is None
        c = lambda x: np.cos(np.pi * x)
        c1 = lambda x: np.cos(np.pi * x) + 1.0
        pi4 = np.pi ** 4
        load = F.SumOfSeparableLoads([([c, c1], pi4), ([c, c], 2 * pi4), ([c1, c], pi4)])
        u = t.Function(spline.V)
        U = spline.solveLinearVariationalProblem(F.Equation(F.BiharmonicForm(), load), u)
        X = spline.V.grids[0].coordinates()
        exact = (np.cos(np.pi * X[:, 0]) + 1.0) * (np.cos(np.pi * X[:, 1]) + 1.0)
        errs[nel] = np.max(np.abs(u.vector().get_local() - exact))
        if nel == 16:
            s = O.BSpline([p, p], [O.uniform_knots(p, -1., 1., nel)] * 2)
            Mo = O.generate_M_tensor(s)
            Ao = O.biharmonic_fe_system_2d(s)
            zo = list(spline.zeroDofs)
            Ko = O.extract_matrix(Mo, Ao, zo)
            bo = spline.assembleVector(load).get_local()
            Uo = spla.spsolve(Ko.tocsc(), bo)
            assert np.linalg.norm(U.get_local() - Uo) <= 1e-9 * np.linalg.norm(Uo)     # direct vs direct
        if nel == 256:
            # cond(K) ~ h^-4: the forward error of ANY direct solve is round-off dominated here (~cond * eps); the
            # backward error is what the factorisation controls
            K = spline.assembleMatrix(F.BiharmonicForm())
            rhs = spline.assembleVector(load)
            r = K.mult(U)
            r.axpy(-1.0, rhs)
            Kn = float(np.sqrt((K.to_scipy().data ** 2).sum()))
            assert r.norm() <= 1e-12 * Kn * U.norm()           # normwise backward error
            assert K.shape[0] == 67600
    assert errs[64] < errs[16] / 50.0 and errs[256] < 1e-3


def test_default_solver_reorders_field_major_systems_and_hands_large_ones_to_cholesky_or_gmres():
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, common as tc, device as dev
    # three fields numbered field after field: bandwidth ~ 2 n/3 as numbered, small after reverse Cuthill-McKee
    p, nel = 2, 24
    kv = [B.uniformKnots(p, 0., 1., nel)] * 2
    gen = t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh([p, p], kv))
    spline = t.ExtractedSpline(gen, 2 * p)
    s = O.BSpline([p, p], [O.uniform_knots(p, 0., 1., nel)] * 2)
    A1, _, _, _ = O.poisson_fe_system(s)
    n1 = A1.shape[0]
    rng = np.random.default_rng(0)
    pat = (abs(A1) > 0).astype(np.float64).tocsr()
    blocks = [[None] * 3 for _ in range(3)]
    for a in range(3):
        for b_ in range(3):
            Bk = pat.copy()
            Bk.data = 0.05 * rng.standard_normal(Bk.nnz)
            blocks[a][b_] = Bk
        blocks[a][a] = blocks[a][a] + sp.identity(n1) * 4.0
    A = sp.bmat(blocks, format="csr")
    K = spline.extractMatrix(A)
    bvec = rng.standard_normal(A.shape[0])
    rhs = spline.extractVector(bvec)
    lu = t.PETScLUSolver()
    lu.parameters["reorder"] = True
    x = dev.DeviceVector(K.shape[0])
    lu.solve(K, x, rhs)
    assert lu.last["reordered"] and lu.last["kl"] < K.shape[0] // 6
    ref = spla.spsolve(K.to_scipy().tocsc(), rhs.get_local())
    assert np.linalg.norm(x.get_local() - ref) <= 1e-10 * np.linalg.norm(ref)
    # a 3-D system beyond the direct solver's budget goes to Jacobi-GMRES (and says so in `last`)
    p, nel = 2, 48
    kv = [B.uniformKnots(p, 0., 1., nel)] * 3
    gen3 = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
    sp0 = gen3.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen3.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spl3 = t.ExtractedSpline(gen3, 2 * p)
    f1 = lambda x_: np.sin(np.pi * x_)
    K3, b3 = spl3.assembleLinearSystem(F.LaplaceForm(), F.SeparableLoadForm([f1] * 3, scale=3 * np.pi ** 2))
    # (round 6: the system is symmetric positive definite -- the banded Cholesky factorisation takes it, beyond the LU's budget)
    d = tc._default_linear_solver()
    x3 = dev.DeviceVector(K3.shape[0])
    d.solve(K3, x3, b3)
    assert d.last["solver"] == "lu" and d.last["factorisation"] == "cholesky"
    r = K3.mult(x3)
    r.axpy(-1.0, b3)
    assert r.norm() <= 1e-9 * b3.norm()
    # the same system with a convection-like skew part is not symmetric: Jacobi-GMRES, as before
    S3 = K3.to_scipy().tocsr()
    up = sp.triu(S3, k=1).tocsr()
    N3 = (S3 + 0.05 * up - 0.05 * up.T).tocsr()
    N3.sort_indices()
    Kn = dev.DeviceCSR.from_scipy(N3)
    d = tc._default_linear_solver()
    xn = dev.DeviceVector(Kn.shape[0])
    d.solve(Kn, xn, b3)
    assert d.last["solver"] == "gmres" and d.last["status"] == 0
    r = Kn.mult(xn)
    r.axpy(-1.0, b3)
    assert r.norm() <= 1e-9 * b3.norm()


def test_default_solver_tries_the_rcm_band_before_giving_lu_up():
    """linearSolver=None on a field-major three-field system whose band as numbered (~2n/3) is far beyond the direct
    solver's budget: the reverse Cuthill-McKee band is evaluated before falling back to Jacobi-GMRES, and the system is
    solved directly like the reference's default LU would; also: a non-zero initial guess set in u reaches the
    Krylov solver through solveLinearSystem (tIGAr/common.py:1250-1254), and a side-dof list edited in place is honoured."""
    import tigar_amd as t
    from tigar_amd import BSplines as B, common as tc, device as dev
    p, nel, nF = 2, 120, 3
    kv = [B.uniformKnots(p, 0., 1., nel)] * 2
    gen = t.EqualOrderSpline(nF, B.ExplicitBSplineControlMesh([p, p], kv))
    sp0 = gen.getScalarSpline(0)
    dofs = sp0.getSideDofs(0, 0)
    ref_sorted = sorted(dofs, reverse=True)
    dofs.sort(reverse=True)                                  # edited in place, same length: the list must win
    gen.addZeroDofs(0, dofs)
    assert gen.zeroDofsArray().tolist() == ref_sorted
    for f in (1, 2):
        gen.addZeroDofs(f, gen.getScalarSpline(f).getSideDofs(0, 0))
    spline = t.ExtractedSpline(gen, 2 * p)
    pat = __import__("tigar_amd.forms", fromlist=["x"]).LaplaceForm().assemble_matrix(
        t.TensorFunctionSpace([sp0.generateMesh(degree=p)], "Lagrange")).to_scipy().tocsr()
    rng = np.random.default_rng(1)
    blocks = [[None] * nF for _ in range(nF)]
    for a in range(nF):
        for b_ in range(nF):
            Bk = pat.copy()
            Bk.data = 0.05 * rng.standard_normal(Bk.nnz)
            blocks[a][b_] = Bk if a != b_ else (Bk + 4.0 * sp.identity(pat.shape[0], format="csr")).tocsr()
    A = sp.bmat(blocks, format="csr")
    K = spline.extractMatrix(A)
    rhs = spline.extractVector(rng.standard_normal(A.shape[0]))
    kl, ku, nb = dev.lu_band_info(K)
    assert nb > 8 * 2 ** 30                                   # as numbered: beyond the budget
    d = tc._default_linear_solver()
    x = dev.DeviceVector(K.shape[0])
    d.solve(K, x, rhs)
    assert d.last["solver"] == "lu" and d.last["reordered"] and d.last["band_bytes"] < 4 * 2 ** 30
    r = K.mult(x)
    r.axpy(-1.0, rhs)
    assert r.norm() <= 1e-10 * rhs.norm()
    # non-zero initial guess through the public solve path: the solver must start from M^T u as the reference does
    solver = t.PETScKrylovSolver("gmres", "jacobi")
    solver.parameters.update({"relative_tolerance": 1e-12, "maximum_iterations": 2, "error_on_nonconvergence": False,
                              "nonzero_initial_guess": True})
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    u.vector().set_local(rng.standard_normal(spline.V.dim()))
    x0 = spline.M.mult_transpose(u.vector())
    want = dev.DeviceVector(data=x0.get_local())
    dev.krylov_solve(K, rhs, want, "gmres", "jacobi", rtol=1e-12, maxit=2, nonzero_initial_guess=True)
    U = spline.solveLinearSystem(K, rhs, u)
    assert solver.last["iterations"] == 2
    assert np.array_equal(U.get_local(), want.get_local())


# ---- round 6: symmetric positive definite systems are factorised as L L^T (csrc/tg_chol.hip) ------------------------------
def _spd_band(rng, n, kl):
    """symmetric band matrix with random off-diagonals, positive definite by diagonal dominance (condition ~ 10)"""
    offs = list(range(1, kl + 1))
    vals = [rng.standard_normal(n - o) for o in offs]
    B = sp.diags(vals + vals, offs + [-o for o in offs], shape=(n, n), format="csr")
    d = np.asarray(abs(B).sum(axis=1)).ravel() * (1.1 + rng.random(n))
    A = (B + sp.diags(d)).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("group", [0, 2, 3, 4])
@pytest.mark.parametrize("n,kl", [(64, 8), (257, 31), (1000, 40), (3001, 130), (5000, 333), (1300, 1100), (2337, 500)])
def test_banded_cholesky_of_spd_systems(n, kl, group, monkeypatch):
    """band widths below, at and above the block width and the tile width, a last block of one column, a band wider than most
    of the matrix, an odd number of blocks with a short last one: the solve runs on the Cholesky path and agrees with SuperLU and
    with the LU path; `group`: that many panels per pass over the trailing triangle (bands of half-width >= 2048 get two, >= 4096
    four)"""
    from tigar_amd import device as dev
    if group:
        monkeypatch.setenv("TIGAR_CHOL_GROUP", str(group))
    if group == 0 and kl % 2:
        monkeypatch.setenv("TIGAR_CHOL_FUSED", "0")          # (a panel and an update kernel per block)
    rng = np.random.default_rng(n + kl)
    A = _spd_band(rng, n, kl)
    xs = rng.standard_normal(n)
    b = A @ xs
    K = dev.DeviceCSR.from_scipy(A)
    c0 = dev.prof_get(8)[1]
    x = dev.DeviceVector(n)
    assert dev.lu_solve(K, dev.DeviceVector(data=b), x) == 0
    assert dev.prof_get(8)[1] == c0 + 1, "the Cholesky path did not run"
    ref = spla.spsolve(A.tocsc(), b)
    res = np.linalg.norm(A @ x.get_local() - b) / np.linalg.norm(b)
    res_ref = np.linalg.norm(A @ ref - b) / np.linalg.norm(b)
    assert res < 1e-10 and res <= 100 * max(res_ref, 1e-16), (res, res_ref)
    monkeypatch.setenv("TIGAR_LU_CHOLESKY", "0")
    x2 = dev.DeviceVector(n)
    assert dev.lu_solve(K, dev.DeviceVector(data=b), x2) == 0
    assert dev.prof_get(8)[1] == c0 + 1
    res2 = np.linalg.norm(A @ x2.get_local() - b) / np.linalg.norm(b)
    assert res <= 100 * max(res2, 1e-16)                    # (as small a backward error as the LU's)


@pytest.mark.parametrize("wgs", ["2", "3", "13", "64"])
def test_cholesky_substitutions_on_several_workgroups(wgs, monkeypatch, capfd):
    """the substitution sweeps with 2 .. 64 workgroups owning the blocks of 32 cyclically (a last block of 9 columns, a band of 21.9
    blocks) give what the single-workgroup substitutions give, and none of their waits gives up"""
    from tigar_amd import device as dev
    rng = np.random.default_rng(77)
    n, kl = 9001, 700
    A = _spd_band(rng, n, kl)
    b = A @ rng.standard_normal(n)
    K = dev.DeviceCSR.from_scipy(A)
    monkeypatch.setenv("TIGAR_CHOL_SWEEP", "0")
    x1 = dev.DeviceVector(n)
    assert dev.lu_solve(K, dev.DeviceVector(data=b), x1) == 0
    monkeypatch.setenv("TIGAR_CHOL_SWEEP", "1")
    monkeypatch.setenv("TIGAR_CHOL_SWEEP_WGS", wgs)
    monkeypatch.setenv("TIGAR_TRACE", "1")
    capfd.readouterr()
    c0 = dev.prof_get(8)[1]
    x2 = dev.DeviceVector(n)
    assert dev.lu_solve(K, dev.DeviceVector(data=b), x2) == 0
    err = capfd.readouterr().err
    assert dev.prof_get(8)[1] == c0 + 1
    assert "%s workgroups" % wgs in err and "gave up" not in err, err
    assert np.max(np.abs(x2.get_local() - x1.get_local())) <= 1e-12 * np.max(np.abs(x1.get_local()))
    assert np.linalg.norm(A @ x2.get_local() - b) <= 1e-12 * np.linalg.norm(b)


def test_cholesky_is_left_for_the_lu_when_the_premise_fails():
    """a symmetric indefinite matrix (a pivot is not positive), a matrix that is not symmetric, one whose pattern is: the LU"""
    from tigar_amd import device as dev
    rng = np.random.default_rng(5)
    n, kl = 900, 25
    A = _spd_band(rng, n, kl)
    xs = rng.standard_normal(n)
    ind = (A - sp.diags(np.where(np.arange(n) == 400, 2.0 * A.diagonal(), 0.0))).tocsr()        # one negative diagonal entry: indefinite
    asym = A.copy().tolil()
    asym[10, 12] += 0.5
    asym = asym.tocsr()
    pat = A.copy().tolil()
    pat[20, 21] = 0.0
    pat = pat.tocsr()
    pat.eliminate_zeros()
    for M in (ind, asym, pat):
        M.sort_indices()
        b = M @ xs
        c0 = dev.prof_get(8)[1]
        x = dev.DeviceVector(n)
        assert dev.lu_solve(dev.DeviceCSR.from_scipy(M), dev.DeviceVector(data=b), x) == 0
        assert dev.prof_get(8)[1] == c0
        assert np.linalg.norm(M @ x.get_local() - b) <= 1e-9 * np.linalg.norm(b)
