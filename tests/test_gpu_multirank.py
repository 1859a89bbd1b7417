"""The multi-rank hot path on real hardware: z-slab row blocks of K per rank, halo exchange + fused 3-scalar
all-reduce per CG iteration (tg_cg / tg_gmres with a communicator), rank-local prolongation -- against the
single-rank run through the same public API.

Two communicators over the same solver code:
* host-staged (tg_comm_create_host + the TCP transport of tigar_amd/launch.py): two ranks SHARING one GPU, so it
  runs on the 1-GPU test box;
* RCCL (ncclSend/ncclRecv/ncclAllReduce over xGMI): needs >= 2 visible GPUs, skipped otherwise.
The reference's counterpart is PETSc's row-block MatPtAP / KSP with VecScatter ghost updates
(tIGAr/common.py:1194-1195, 1255-1261)."""
import os
import sys
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(d, p, nel, method, periodic0=False, explicit=False, nels=None, periodic=None):
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, common as tc
    nels = [nel] * d if nels is None else list(nels)
    per = set(periodic) if periodic is not None else ({0} if periodic0 else set())
    kv = [B.uniformKnots(p, 0., 1., nels[k], k in per) for k in range(d)]
    gen = t.EqualOrderSpline(tc.selfcomm, 1, B.ExplicitBSplineControlMesh([p] * d, kv))
    sp0 = gen.getScalarSpline(0)
    for direction in range(d):
        if direction not in per:
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p, comm=tc.selfcomm)
    if explicit:
        A = F.LaplaceForm().assemble_matrix(spline.V).to_scipy().tolil()
        A[5, A.shape[1] - 7] = 0.25
        K = spline.extractMatrix(A.tocsr(), diag=1.5)
    else:
        K = spline.assembleMatrix(F.LaplaceForm(), diag=1.5)
    f1 = lambda x: np.sin(np.pi * x)
    rhs = spline.assembleVector(F.SeparableLoadForm([f1] * d, scale=d * np.pi ** 2))
    solver = t.PETScKrylovSolver(*(method.split(":") if ":" in method else (method, "jacobi")))
    solver.parameters["relative_tolerance"] = 1e-10
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    from tigar_amd.device import DeviceVector
    return (K.to_scipy(), rhs.get_local(), U.get_local(), u.vector().get_local(), solver.last["iterations"],
            gen.cpFuncs[0].vector().get_local(), spline.M.mult_transpose(u.vector()).get_local(),
            lambda v: spline.M.mult_transpose(DeviceVector(data=v)).get_local())


def _run_ranks(tmp_path, world, kind, d, p, nel, method, port, env_more=None):
    from tigar_amd.launch import spawn_local
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), "TIGAR_COMM": kind}
    env.update(env_more or {})
    if kind in ("host", "ipc"):
        env["TIGAR_DEVICE"] = "0"                       # every rank on the one GPU
    rc = spawn_local(world, [os.path.join(ROOT, "tests", "gpu_rank_worker.py"), str(tmp_path), str(d), str(p),
                             str(nel), method], env_extra=env, port=port)
    assert rc == 0, "a rank failed"
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


def _compare(parts, ref, world, kind, its_slack=1):
    Ks, rhs, U, u, its, cp0, MTu = ref[:7]
    Ks = Ks.tocsr()
    # M^T of the FE function the RANKS computed (their rows put together): what their initial guesses must equal
    u_ranks = np.zeros_like(u)
    for z in parts:
        u_ranks[int(z["g"][2]):int(z["g"][3])] = z["u"]
    MTu = ref[7](u_ranks) if len(ref) > 7 else MTu
    dof_cover = np.zeros(Ks.shape[0], dtype=int)
    fe_cover = np.zeros(u.shape[0], dtype=int)
    for r, z in enumerate(parts):
        g0, g1, r0, r1 = [int(v) for v in z["g"]]
        assert list(z["comm"]) == [r, world, ("rccl", "host", "ipc").index(kind)]   # what the communicator itself reports
        Kl = sp.csr_matrix((z["K_data"], z["K_indices"], z["K_indptr"]), shape=(g1 - g0, Ks.shape[1]))
        Kr = Ks[g0:g1]
        assert np.array_equal(Kl.indptr, Kr.indptr) and np.array_equal(Kl.indices, Kr.indices)   # pattern identical
        assert abs(Kl - Kr).max() <= 1e-12 * abs(Ks).max()
        assert np.max(np.abs(z["rhs"] - rhs[g0:g1])) <= 1e-13 * np.max(np.abs(rhs))
        assert np.max(np.abs(z["U"] - U[g0:g1])) <= 1e-8 * np.max(np.abs(U))
        if r1 > r0:                                  # (a rank of a thin slab may own no FE rows)
            assert np.max(np.abs(z["u"] - u[r0:r1])) <= 1e-8 * np.max(np.abs(u))
            assert np.max(np.abs(z["cp0"] - cp0[r0:r1])) <= 1e-14
        # same Krylov iteration count as the single-rank solve
        assert abs(int(z["its"][0]) - its) <= its_slack, "iterations %d on %d ranks, %d on one" % (int(z["its"][0]), world, its)
        assert int(z["its"][1]) <= 2                     # restart from the solution: (almost) converged at once
        # the initial guess solveLinearSystem takes from u (M^T u, tIGAr/common.py:1250-1254): every contribution there,
        # also for the dofs next to a slab boundary (ghost rows of u from the z-neighbours)
        assert np.max(np.abs(z["guess"] - MTu[g0:g1])) <= 1e-12 * np.max(np.abs(MTu))
        assert np.max(np.abs(z["U2"] - z["U"])) <= 1e-8 * np.max(np.abs(U))
        dof_cover[g0:g1] += 1
        fe_cover[r0:r1] += 1
    assert np.all(dof_cover == 1) and np.all(fe_cover == 1)       # rows partitioned exactly


@pytest.mark.parametrize("d,p,nel,method,world", [(3, 2, 20, "cg", 2), (3, 3, 14, "cg", 3), (3, 2, 12, "gmres", 2)])
def test_host_staged_ranks_sharing_one_gpu(tmp_path, d, p, nel, method, world):
    ref = _single(d, p, nel, method)
    parts = _run_ranks(tmp_path, world, "host", d, p, nel, method, 29500 + 37 * (d * 100 + p * 10 + world))
    _compare(parts, ref, world, "host")


@pytest.mark.parametrize("kind,d,p,nel,world", [("host", 3, 2, 24, 2), ("ipc", 3, 3, 20, 2), ("ipc", 3, 2, 24, 3)])
def test_half_storage_product_on_several_ranks(tmp_path, kind, d, p, nel, world):
    """CG on z slabs with every rank multiplying by the half-storage copy of ITS rows (csrc/tg_symgrid.hip: entries above
    the slab gathered, not scattered; entries below it taken from the CSR rows of the first planes) after the halo exchange
    of the direction vector: the checks of the sliced copy, and the same iteration count"""
    ref = _single(d, p, nel, "cg")
    parts = _run_ranks(tmp_path, world, kind, d, p, nel, "cg", 30700 + 41 * (p * 10 + world) + (kind == "ipc"),
                       {"TIGAR_SPMV_SYM": "2", "TIGAR_KSP_PERSISTENT": "0"})
    _compare(parts, ref, world, kind)
    for z in parts:
        assert int(z["symgrid"][0]) >= 1
        assert int(z["overlapped"][0]) >= int(z["its"][0])     # (all z chunks but the last beside the halo exchange)


def test_products_beside_the_halo_exchange_change_nothing(tmp_path):
    """CG computes the rows without halo columns while the halo of the direction vector travels (tg_comm_halo_begin /
    _end on the communicator's stream); every row is summed by the same kernel in the same order either way, so the
    iterates are bit-identical to those with the exchange in front of the product (TIGAR_CG_OVERLAP=0)."""
    d, p, nel, world = 3, 2, 24, 3
    a, b = tmp_path / "on", tmp_path / "off"
    a.mkdir(), b.mkdir()
    on = _run_ranks(a, world, "host", d, p, nel, "cg", 31337)
    off = _run_ranks(b, world, "host", d, p, nel, "cg", 31737, {"TIGAR_CG_OVERLAP": "0"})
    for r in range(world):
        # every product of the solve (the host runs up to two iterations ahead of the one it has seen converge)
        assert int(on[r]["its"][0]) + 1 <= int(on[r]["overlapped"][0]) <= int(on[r]["its"][0]) + 3
        assert int(off[r]["overlapped"][0]) == 0
        assert int(on[r]["its"][0]) == int(off[r]["its"][0])
        assert np.array_equal(on[r]["U"], off[r]["U"])


@pytest.mark.parametrize("d,p,nel,method,world", [(3, 2, 20, "cg", 2), (3, 3, 14, "cg", 3), (3, 2, 12, "gmres", 2)])
def test_ipc_ranks_sharing_one_gpu(tmp_path, d, p, nel, method, world):
    """The IPC communicator (device mailboxes through hipIpcOpenMemHandle + flags in shared memory, waits inside the
    kernels): the same checks as the host-staged one, and no host wait inside any exchange of the CG solve."""
    ref = _single(d, p, nel, method)
    parts = _run_ranks(tmp_path, world, "ipc", d, p, nel, method, 30100 + 37 * (d * 100 + p * 10 + world))
    _compare(parts, ref, world, "ipc")
    for z in parts:
        assert int(z["host_waits"][0]) == 0


def test_ipc_iterates_are_those_of_the_host_staged_run(tmp_path):
    """The enqueue-only CG (iterations enqueued two ahead of the norm the host has seen, products past convergence
    gated on the device, halo beside the interior rows) over the IPC communicator gives bit for bit the iterate of the
    host-staged run, whose every exchange waits for the host: both reductions add the ranks' contributions in rank
    order, the halo carries the same doubles, and the two iterations enqueued past convergence change nothing
    (they used to: the frozen update handed the already reduced scalars to the next all-reduce on EVERY rank)."""
    d, p, nel, world = 3, 2, 24, 3
    a, b = tmp_path / "ipc", tmp_path / "host"
    a.mkdir(), b.mkdir()
    ipc = _run_ranks(a, world, "ipc", d, p, nel, "cg", 32137)
    host = _run_ranks(b, world, "host", d, p, nel, "cg", 32537)
    for r in range(world):
        assert int(ipc[r]["its"][0]) == int(host[r]["its"][0])
        assert np.array_equal(ipc[r]["U"], host[r]["U"])
        assert np.array_equal(ipc[r]["u"], host[r]["u"])
        assert ipc[r]["resnorm"][0] == host[r]["resnorm"][0]
        assert int(ipc[r]["host_waits"][0]) == 0 and int(host[r]["host_waits"][0]) > 0
        assert int(ipc[r]["overlapped"][0]) >= int(ipc[r]["its"][0]) + 1      # products beside the exchange


@pytest.mark.parametrize("how", ["device", "scipy"])
def test_explicit_fe_matrix_with_several_ranks(tmp_path, how):
    """extractMatrix(A) with an ASSEMBLED matrix (here with a coupling added by hand, outside the element-coupling
    pattern) on several ranks: every rank cuts the row blocks of its slab out of its copy; K rows and the solution equal
    the single-rank product (tIGAr/common.py:1194-1195 applies MatPtAP to whatever distributed A it is given)."""
    d, p, nel, world = 3, 2, 10, 2
    ref = _single(d, p, nel, "gmres", explicit=True)
    parts = _run_ranks(tmp_path, world, "ipc", d, p, nel, "gmres", 34411 + len(how), {"TIGAR_TEST_EXPLICIT_A": how})
    _compare(parts, ref, world, "ipc")


def test_patch_periodic_across_the_slabs_with_several_ranks(tmp_path):
    """a patch that is periodic in x: only the slab direction (the last one) needs an open knot vector"""
    d, p, nel, world = 3, 2, 9, 3
    ref = _single(d, p, nel, "cg", periodic0=True)
    parts = _run_ranks(tmp_path, world, "ipc", d, p, nel, "cg", 34611, {"TIGAR_TEST_PERIODIC0": "1"})
    _compare(parts, ref, world, "ipc")
    # every rank's rows of K came out of the tensor line walks (on the unwrapped space, then folded: kronptap.unwrapped)
    assert all(int(q["tensor_walks"][0]) > 0 for q in parts)


@pytest.mark.parametrize("p,nels,periodic,world,method", [
    (2, (5, 11, 13), (), 3, "cg"),                 # different element counts per direction
    (3, (9, 4, 12), (0,), 2, "cg"),                # few elements across, periodic in x
    (2, (6, 7, 9), (0, 1), 3, "gmres"),            # periodic in both directions of the planes
    (3, (4, 5, 7), (1,), 2, "bicgstab"),           # slabs of 5 dof planes each
    (1, (7, 6, 8), (), 3, "cg"),                   # trilinear
    (2, (8, 9, 8), (), 3, "cg"),                   # slabs thinner than the FE rows a rank's forms read: ghost rows from
                                                   # beyond the neighbour (two sweeps along the chain of ranks)
    (2, (5, 5, 6), (), 3, "cg"),                   # 8 dof planes on 3 ranks: the last rank owns no FE rows
    (1, (6, 5, 3), (1,), 3, "gmres"),              # 4 dof planes on 3 ranks
])
def test_unequal_directions_on_several_ranks(tmp_path, p, nels, periodic, world, method):
    """element counts that differ per direction, periodic directions other than the slab direction in every combination,
    thin slabs, p = 1: the rank-local rows of K, M^T b, the solution and the prolongation equal the single-rank run
    (tIGAr/common.py:1194-1195, 1255-1261 on PETSc's row blocks)"""
    d = 3
    ref = _single(d, p, nels[0], method, nels=nels, periodic=periodic)
    env = {"TIGAR_TEST_NELS": ",".join(str(n) for n in nels), "TIGAR_TEST_PERIODIC": "".join(str(k) for k in periodic)}
    parts = _run_ranks(tmp_path, world, "ipc", d, p, nels[0], method, 36011 + 13 * (p * 10 + world + len(periodic)), env)
    # (BiCGStab's recurrences amplify the last bits of the dot products, which several ranks sum in another order:
    #  the iteration count may move by a few, the solution is held to the same 1e-8)
    _compare(parts, ref, world, "ipc", its_slack=6 if method == "bicgstab" else 1)


@pytest.mark.parametrize("case,world,kind", [("shell2d", 2, "ipc"), ("elasticity3d", 3, "ipc"), ("shell2d", 3, "host"),
                                             ("mapped_elasticity3d", 2, "ipc"), ("mapped_elasticity3d", 3, "ipc")])
def test_several_fields_on_several_ranks(tmp_path, case, world, kind):
    _several_fields(tmp_path, case, world, kind, {})


@pytest.mark.parametrize("case,world,p,nels,periodic", [
    ("elasticity3d", 2, 1, (5, 3, 8), ""),            # trilinear, different element counts
    ("elasticity3d", 3, 2, (3, 6, 9), "0"),           # periodic in x
    ("elasticity3d", 2, 3, (4, 4, 6), "01"),          # cubic, periodic in x and y
    ("shell2d", 3, 2, (9, 12), "0"),                  # 2-D three fields, periodic in x
    ("shell2d", 2, 4, (6, 9), ""),                    # quartic
])
def test_several_fields_on_several_ranks_other_shapes(tmp_path, case, world, p, nels, periodic):
    """the same checks on patches with other degrees, element counts per direction and periodic directions (found by hand
    after the random single-field multi-rank runs: this combination had no coverage)"""
    env = {"TIGAR_TEST_FP": str(p), "TIGAR_TEST_FNELS": ",".join(str(n) for n in nels), "TIGAR_TEST_FPER": periodic}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        _several_fields(tmp_path, case, world, "ipc", env)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("world,kind,degs,nels", [(2, "ipc", "1,1,1", "5,4,9"), (3, "host", "1,1,1", "3,4,8"), (2, "ipc", "2,2,2", "3,3,6"),
                                                  (3, "ipc", "1,1,1", "4,3,4")])
def test_fields_on_different_bases_on_several_ranks(tmp_path, world, kind, degs, nels):
    """BSplineCompat("RT") -- the space of the reference's Krylov + MPI demos (demos/taylor-green/taylor-green-3d.py:42-90;
    tIGAr/compatibleSplines.py:21-101) -- in z-slabs over 2-3 ranks: one split of the plane index for the three fields (they
    have different numbers of planes), the dofs interleaved plane by plane, every block K_fg = M_f^T A_fg M_g by the line
    walks with different row / column bases; rows of K, M^T b, the GMRES solution and u = M U of every rank against the
    single-rank resident run (which the kernel- and API-level tests pin to the oracle)."""
    env = {"TIGAR_TEST_FDEGS": degs, "TIGAR_TEST_FNELS": nels}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        _several_fields(tmp_path, "rt3d", world, kind, env)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fields_on_different_bases_streamed_through_one_gpu():
    """the same engine on ONE rank with the operator kept implicit and the patch streamed in sub-slabs of two dof planes:
    K and M^T b equal the resident path's after undoing the plane-wise interleaving, the walks ran for all nine blocks, and
    the solution / prolongation agree"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_rank_worker_fields as W
    import tigar_amd as t
    from tigar_amd import common as tc, device as dev
    gen, spline, K, rhs, method = W.problem("rt3d", tc.selfcomm)
    Kref, rref = K.to_scipy().tocsr(), rhs.get_local()
    os.environ["TIGAR_IMPLICIT_M"] = "1"
    os.environ["TIGAR_SUB_PLANES"] = "2"
    try:
        dev.prof_reset()
        gen2, spline2, K2, rhs2, _ = W.problem("rt3d", tc.selfcomm)
        assert getattr(gen2.M, "is_implicit", False) and gen2.M.nfields == 3
        assert dev.prof_get(5)[1] >= 9 * 3                      # nine blocks, several sub-slabs each
        dofs = spline2.localDofIndices()
        n2o = spline2._slab_path().new_of_old()
        n = Kref.shape[0]
        old_of_new = np.empty(n, dtype=np.int64)
        old_of_new[n2o] = np.arange(n)
        K2s = K2.to_scipy().tocsr()
        Kr = Kref[dofs][:, old_of_new].tocsr()
        Kr.sort_indices()
        K2s.sort_indices()
        assert np.array_equal(K2s.indptr, Kr.indptr) and np.array_equal(K2s.indices, Kr.indices)
        assert abs(K2s - Kr).max() <= 1e-12 * abs(Kref).max()
        assert np.max(np.abs(rhs2.get_local() - rref[dofs])) <= 1e-13 * np.max(np.abs(rref))
        solver = t.PETScKrylovSolver("gmres", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-11
        for s_ in (spline, spline2):
            s_.setSolverOptions(linearSolver=solver)
        u, u2 = t.Function(spline.V), t.Function(spline2.V, spline2.localFERange())
        U = spline.solveLinearSystem(K, rhs, u).get_local()
        U2 = spline2.solveLinearSystem(K2, rhs2, u2).get_local()
        assert np.max(np.abs(U2 - U[dofs])) <= 1e-8 * np.max(np.abs(U))
        assert np.max(np.abs(u2.vector().get_local() - u.vector().get_local())) <= 1e-8 * np.max(np.abs(u.vector().get_local()))
        # an assembled FE matrix on the same spline goes through the same engine and numbering
        from tigar_amd import forms as F
        K3 = spline2.extractMatrix(F.ElasticityForm(2.0, 1.0).assemble_matrix(spline.V), diag=1.5).to_scipy().tocsr()
        K3.sort_indices()
        assert np.array_equal(K3.indices, Kr.indices) and abs(K3 - Kr).max() <= 1e-12 * abs(Kref).max()
    finally:
        os.environ.pop("TIGAR_IMPLICIT_M", None)
        os.environ.pop("TIGAR_SUB_PLANES", None)


def _several_fields(tmp_path, case, world, kind, env_more):
    """EqualOrderSpline(nFields = 3) split into z-slabs (VERDICT r2 missing #2): a rank owns the dof planes of every field,
    the dofs are interleaved plane by plane so that its rows of K are one contiguous block (localDofIndices() gives the
    reference index of every local dof); K = [M_s^T A_fg M_s] block by block through the scalar slab engine -- an
    assembled 3-field matrix (cfg5's kind) and ElasticityForm (blocks as Kronecker sums, fused into the first pass) --
    against the single-rank resident product, the solution and the prolongation against the single-rank solve."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_rank_worker_fields as W
    import tigar_amd as t
    from tigar_amd import common as tc
    gen, spline, K, rhs, method = W.problem(case, tc.selfcomm)
    assert not getattr(gen.M, "is_implicit", False)
    solver = t.PETScKrylovSolver(*(method.split(":") if ":" in method else (method, "jacobi")))
    solver.parameters["relative_tolerance"] = 1e-10
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u).get_local()
    Kref, rref, uref, its = K.to_scipy().tocsr(), rhs.get_local(), u.vector().get_local(), solver.last["iterations"]
    from tigar_amd.launch import spawn_local
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), "TIGAR_COMM": kind, "TIGAR_DEVICE": "0"}
    env.update(env_more)
    rc = spawn_local(world, [os.path.join(ROOT, "tests", "gpu_rank_worker_fields.py"), str(tmp_path), case], env_extra=env,
                     port=35011 + 13 * world + len(case) + 7 * len(env_more.get("TIGAR_TEST_FNELS", "")))
    assert rc == 0
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    n = Kref.shape[0]
    cover = np.zeros(n, dtype=int)
    ucover = np.zeros(uref.size, dtype=int)
    for r, z in enumerate(parts):
        dofs, n2o = z["dofs"], z["new_of_old"]
        g0, g1 = [int(v) for v in z["g"]]
        assert g1 - g0 == dofs.size and np.array_equal(n2o[dofs], np.arange(g0, g1))     # local block contiguous
        old_of_new = np.empty(n, dtype=np.int64)
        old_of_new[n2o] = np.arange(n)
        Kl = sp.csr_matrix((z["K_data"], z["K_indices"], z["K_indptr"]), shape=(dofs.size, n))
        # the reference rows of these dofs, columns renamed into the distributed numbering
        Kr = Kref[dofs][:, old_of_new].tocsr()
        Kr.sort_indices()
        Kl.sort_indices()
        assert np.array_equal(Kl.indptr, Kr.indptr) and np.array_equal(Kl.indices, Kr.indices)   # pattern identical
        assert abs(Kl - Kr).max() <= 1e-12 * abs(Kref).max()
        assert np.max(np.abs(z["rhs"] - rref[dofs])) <= 1e-13 * np.max(np.abs(rref))
        assert np.max(np.abs(z["U"] - U[dofs])) <= 1e-8 * np.max(np.abs(U))
        rows = np.concatenate([np.arange(a, b) for a, b in z["fe"]])
        if rows.size:                                # (a rank of a thin slab may own no FE rows)
            assert np.max(np.abs(z["u"] - uref[rows])) <= 1e-8 * np.max(np.abs(uref))
        assert abs(int(z["its"][0]) - its) <= max(1, its // 20)
        if kind == "ipc":
            assert int(z["host_waits"][0]) == 0 and int(z["kind"][0]) == 2
        cover[dofs] += 1
        ucover[rows] += 1
    assert np.all(cover == 1) and np.all(ucover == 1)


def test_several_fields_streamed_through_one_gpu():
    """the same field-block engine on ONE rank with the operator kept implicit (TIGAR_IMPLICIT_M=1) and the patch streamed
    in sub-slabs of three dof planes: K, M^T b, the solution and u = M U equal the resident path's after undoing the
    plane-wise interleaving of the fields"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_rank_worker_fields as W
    import tigar_amd as t
    from tigar_amd import common as tc
    gen, spline, K, rhs, method = W.problem("elasticity3d", tc.selfcomm)
    Kref, rref = K.to_scipy().tocsr(), rhs.get_local()
    os.environ["TIGAR_IMPLICIT_M"] = "1"
    os.environ["TIGAR_SUB_PLANES"] = "3"
    try:
        gen2, spline2, K2, rhs2, _ = W.problem("elasticity3d", tc.selfcomm)
        assert getattr(gen2.M, "is_implicit", False) and len(spline2._slab_path().scalar.sub_slabs()) >= 3
        dofs = spline2.localDofIndices()
        n2o = spline2._slab_path().new_of_old()
        n = Kref.shape[0]
        old_of_new = np.empty(n, dtype=np.int64)
        old_of_new[n2o] = np.arange(n)
        K2s = K2.to_scipy().tocsr()
        Kr = Kref[dofs][:, old_of_new].tocsr()
        Kr.sort_indices()
        K2s.sort_indices()
        assert np.array_equal(K2s.indptr, Kr.indptr) and np.array_equal(K2s.indices, Kr.indices)
        assert abs(K2s - Kr).max() <= 1e-12 * abs(Kref).max()
        assert np.max(np.abs(rhs2.get_local() - rref[dofs])) <= 1e-13 * np.max(np.abs(rref))
        solver = t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-11
        for s_, K_, r_ in ((spline, K, rhs), (spline2, K2, rhs2)):
            s_.setSolverOptions(linearSolver=solver)
        u, u2 = t.Function(spline.V), t.Function(spline2.V, spline2.localFERange())
        U = spline.solveLinearSystem(K, rhs, u).get_local()
        U2 = spline2.solveLinearSystem(K2, rhs2, u2).get_local()
        assert np.max(np.abs(U2 - U[dofs])) <= 1e-8 * np.max(np.abs(U))
        assert np.max(np.abs(u2.vector().get_local() - u.vector().get_local())) <= 1e-8 * np.max(np.abs(u.vector().get_local()))
        # an EXPLICIT FE matrix on the same spline (extractMatrix kept the field-major order for it while
        # extractVector / solveLinearSystem used the plane-wise one -- K and M^T b permuted differently, silently): an
        # assembled DeviceCSR and a scipy matrix go through the same engine and numbering as the form did
        from tigar_amd import forms as F
        Aex = F.ElasticityForm(2.0, 1.0).assemble_matrix(spline.V)
        for A3 in (Aex, Aex.to_scipy()):
            K3 = spline2.extractMatrix(A3, diag=1.5).to_scipy().tocsr()
            K3.sort_indices()
            assert np.array_equal(K3.indptr, Kr.indptr) and np.array_equal(K3.indices, Kr.indices)
            assert abs(K3 - Kr).max() <= 1e-12 * abs(Kref).max()
        # FEtoIGA (its identity matrix is such an explicit A): the dofs of u = M U come back in the local numbering
        back = spline2.FEtoIGA(u).get_local()
        assert np.max(np.abs(back - U[dofs])) <= 1e-7 * np.max(np.abs(U))
    finally:
        os.environ.pop("TIGAR_IMPLICIT_M", None)
        os.environ.pop("TIGAR_SUB_PLANES", None)


@pytest.mark.parametrize("world,kind,d,p,nel,with_dofs", [(2, "ipc", 2, 2, 12, False), (3, "ipc", 2, 3, 12, True),
                                                          (2, "host", 3, 2, 8, True),
                                                          # slabs of 2-3 dof planes: the forms' ghost rows come from beyond
                                                          # the neighbour (two sweeps along the chain of ranks)
                                                          (3, "ipc", 3, 2, 6, True), (4, "ipc", 2, 2, 7, False),
                                                          (3, "host", 2, 3, 8, True)])
def test_newton_on_several_ranks(tmp_path, capfd, world, kind, d, p, nel, with_dofs):
    """solveNonlinearVariationalProblem with several ranks (VERDICT r3 missing #1; tIGAr/common.py:1304-1348 is MPI-aware:
    ||M^T R|| at :1330 is a global norm, u.assign(u - du) at :1343 works on distributed vectors): u, du and the IGA dofs
    stay rank-local, the forms read u with its ghost rows, and the history of relative norms equals the single-rank
    run's to 1e-12 -- as do the solution and the returned dofs on every rank's rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_rank_worker_newton as W
    from tigar_amd import common as tc
    spline, u, dofs, hist = W.run(tc.selfcomm, d, p, nel, with_dofs)
    uref = u.vector().get_local()
    dref = dofs.get_local() if dofs is not None else None
    capfd.readouterr()
    from tigar_amd.launch import spawn_local
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), "TIGAR_COMM": kind, "TIGAR_DEVICE": "0"}
    rc = spawn_local(world, [os.path.join(ROOT, "tests", "gpu_rank_worker_newton.py"), str(tmp_path), str(d), str(p), str(nel),
                             "1" if with_dofs else "0"], env_extra=env, port=36411 + 17 * world + d)
    assert rc == 0, "a rank failed"
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert len(hist) >= 4                                      # a genuinely nonlinear problem
    cover = np.zeros(uref.size, dtype=int)
    for z in parts:
        g0, g1, r0, r1 = [int(v) for v in z["g"]]
        assert len(z["hist"]) == len(hist)
        for a, b in zip(z["hist"], hist):
            assert abs(a - b) <= 1e-12, (list(z["hist"]), hist)
        assert z["u"].size == r1 - r0                          # rank-local rows only
        # (a rank of a thin slab may own no FE rows at all)
        assert r1 == r0 or np.max(np.abs(z["u"] - uref[r0:r1])) <= 1e-10 * np.max(np.abs(uref))
        if with_dofs:
            assert z["dofs"].size == g1 - g0
            assert np.max(np.abs(z["dofs"] - dref[g0:g1])) <= 1e-10 * np.max(np.abs(dref))
        cover[r0:r1] += 1
        # gatherFunction: the full-length function on every rank, the same bits everywhere
        assert z["u_gathered"].size == uref.size
        assert np.max(np.abs(z["u_gathered"] - uref)) <= 1e-10 * np.max(np.abs(uref))
        assert np.array_equal(z["u_gathered"].view(np.int64), parts[0]["u_gathered"].view(np.int64))
        assert r1 == r0 or np.array_equal(z["u_gathered"][r0:r1], z["u"])
    assert np.all(cover == 1)
    out = capfd.readouterr().out
    assert out.count("Solver iteration: 0 ,") == 1            # rank 0 alone prints the reference's progress line


@pytest.mark.parametrize("method,world", [("cg:chebyshev", 2), ("bicgstab:jacobi", 3)])
def test_chebyshev_cg_and_bicgstab_on_several_ranks(tmp_path, method, world):
    """the solvers of csrc/tg_krylov.hip added in round 4 take the same halo exchange / all-reduce entry points: rows of K,
    M^T b, the solution and the iteration count against the single-rank run"""
    d, p, nel = 3, 2, 14
    ref = _single(d, p, nel, method)
    parts = _run_ranks(tmp_path, world, "ipc", d, p, nel, method, 37011 + world)
    _compare(parts, ref, world, "ipc")


def test_ipc_dead_peer_is_an_error_not_a_hang(tmp_path):
    """A rank that is gone must not leave its peers spinning on the GPU: the waits inside the IPC kernels give up
    after TIGAR_IPC_TIMEOUT_S seconds of wall clock and the next host wait reports it."""
    from tigar_amd.launch import spawn_local
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), "TIGAR_COMM": "ipc", "TIGAR_DEVICE": "0",
           "TIGAR_IPC_TIMEOUT_S": "3"}
    rc = spawn_local(2, [os.path.join(ROOT, "tests", "ipc_dead_peer_worker.py"), str(tmp_path)], env_extra=env, port=34211)
    assert rc == 0
    secs, msg = open(os.path.join(str(tmp_path), "rank0.txt")).read().split("\n")[:2]
    assert "gave up waiting for an all-reduce" in msg
    assert 2.0 <= float(secs) <= 20.0


@pytest.mark.parametrize("kind,world,method", [("host", 3, "cg"), ("ipc", 2, "cg"), ("ipc", 3, "cg"), ("ipc", 2, "gmres"),
                                               ("host", 3, "gmres")])
def test_iterations_enqueued_past_convergence_change_nothing(tmp_path, kind, world, method):
    """The host enqueues CG iterations two ahead of the norm it has read; the ones past convergence must leave x, the
    reported norm and the iteration count exactly as a run that reads every norm first (TIGAR_CG_LOOK=0).  With several
    ranks the frozen update has to hand gamma / nu to the next all-reduce ONCE, not once per rank."""
    d, p, nel = 3, 2, 18
    a, b = tmp_path / "ahead", tmp_path / "lockstep"
    a.mkdir(), b.mkdir()
    ladder = {"TIGAR_TEST_RTOLS": ",".join("%g" % (10.0 ** (-0.5 * k)) for k in range(4, 24))}
    ahead = _run_ranks(a, world, kind, d, p, nel, method, 33137 + world, ladder)
    lock = _run_ranks(b, world, kind, d, p, nel, method, 33537 + world, dict(ladder, TIGAR_CG_LOOK="0"))
    for r in range(world):
        assert int(ahead[r]["its"][0]) == int(lock[r]["its"][0])
        assert np.array_equal(ahead[r]["U"], lock[r]["U"])
        assert ahead[r]["resnorm"][0] == lock[r]["resnorm"][0]
        # twenty tolerances: some solves end with world * ||B r||^2 above tol^2 although ||B r||^2 itself is below
        assert len(ahead[r]["ladder_its"]) == 20
        assert np.array_equal(ahead[r]["ladder_its"], lock[r]["ladder_its"])
        assert np.array_equal(ahead[r]["ladder_res"], lock[r]["ladder_res"])
        assert np.array_equal(ahead[r]["ladder_U"], lock[r]["ladder_U"])


@pytest.mark.parametrize("d,p,nel,method", [(3, 3, 24, "cg"), (3, 2, 16, "gmres")])
def test_rccl_ranks_one_gpu_each(tmp_path, d, p, nel, method):
    from tigar_amd import device as dev
    ndev = dev.device_count()
    if ndev < 2:
        pytest.skip("RCCL needs one GPU per rank; %d visible" % ndev)
    world = min(ndev, 4)
    ref = _single(d, p, nel, method)
    parts = _run_ranks(tmp_path, world, "rccl", d, p, nel, method, 29900 + world)
    _compare(parts, ref, world, "rccl")


def test_bench_launcher_with_two_ranks_on_the_shared_gpu(tmp_path):
    """the first real multi-GPU run must not also be the first run of ``bench.py --gpus N`` itself.  The
    launcher path (bench.py spawns its ranks through tigar_amd.launch.spawn_local, the ranks find each other, rank 0 prints
    ONE JSON line) with two ranks SHARING the test box's GPU over the IPC communicator: devices actually used = 1, the
    communicator says what served, per-rank stage times are there, the self-check of the timed solve passes, and the
    iteration count is that of a one-rank run of the same command."""
    import json
    import subprocess

    def run(gpus, extra_env):
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29500 + 53 * gpus + 7))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        env.update(extra_env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "cfg2",
                              "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--companion", "0", "--live-traffic", "0"],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, "exactly one JSON line: %r" % (out.stdout[-500:],)
        return json.loads(lines[0])

    one = run(1, {})
    two = run(2, {"TIGAR_COMM": "ipc", "TIGAR_DEVICE": "0"})
    cfg = two["config"]
    assert two["n_gpus"] == 1                               # devices actually used, not what argv asked for
    assert cfg["ranks"] == 2 and cfg["communicator"] == "ipc"
    assert cfg["per_rank_stages_s"] is not None and len(cfg["per_rank_stages_s"]) == 2
    assert all(set(("extract", "ptap", "mtb", "solve")) <= set(r) for r in cfg["per_rank_stages_s"])
    assert sum(r["dof_rows"] for r in cfg["per_rank_stages_s"]) == cfg["dofs"] == one["config"]["dofs"]
    assert cfg["self_check_rel_residual_all_ranks"] is not None and cfg["self_check_rel_residual_all_ranks"] <= 1e-6
    assert cfg["cg_iterations"] == one["config"]["cg_iterations"]
    assert two["metric"] == one["metric"] and two["unit"] == "DoF/s" and two["steps"] == 1
    assert "roofline" in two and two["roofline"]["bound"] == "hbm"


def test_rccl_asked_for_on_a_shared_gpu_falls_back_to_ipc(tmp_path):
    """``TIGAR_COMM=rccl`` with two ranks on ONE device -- RCCL cannot form that communicator (it refuses two
    ranks of one GPU, or its initialisation does not return: ``tg_nccl_init_with_timeout``, csrc/tg_dist.hip) -- must end in the
    IPC communicator on every rank, agreed through the transport, with a note that says why, and the run must give what one
    rank gives: the path a first multi-GPU node takes when RCCL is not usable there."""
    import json
    import subprocess

    def run(gpus, extra_env):
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29500 + 61 * gpus + 11))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        env.update(extra_env)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "cfg2",
                              "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--companion", "0", "--live-traffic", "0"],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0]), out.stderr

    one, _ = run(1, {})
    two, err = run(2, {"TIGAR_COMM": "rccl", "TIGAR_DEVICE": "0", "TIGAR_RCCL_TIMEOUT_S": "25", "TIGAR_COMM_SELFTEST_S": "25"})
    cfg = two["config"]
    assert cfg["ranks"] == 2 and cfg["communicator_requested"] == "rccl" and cfg["communicator"] == "ipc"
    assert cfg["communicator_fallback"] and "RCCL" in cfg["communicator_fallback"][0]
    assert "using the IPC communicator" in err
    assert cfg["cg_iterations"] == one["config"]["cg_iterations"]
    assert cfg["self_check_rel_residual_all_ranks"] <= 1e-6
    assert all(r["halo_bytes_per_product"] > 0 for r in cfg["per_rank_stages_s"])
