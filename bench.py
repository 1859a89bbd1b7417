#!/usr/bin/env python3
"""
bench.py -- tIGAr extraction hot path on MI355X: extraction-operator build (generateM) ->
M^T A M + M^T b (extractMatrix / extractVector) -> Krylov solve (solveLinearSystem), on a
synthetic tensor-product B-spline Poisson patch, driven through the
reference's own API surface (EqualOrderSpline -> ExtractedSpline -> assembleMatrix / assembleVector
-> solveLinearSystem) for every workload and rank count.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|auto]

One "step" = one full pass of the hot path over the patch: generator construction (1-D tables, M or
its implicit form, control functions), explicit/implicit M^T, M^T A M with boundary conditions, M^T b,
Jacobi-CG, prolongation u = M U.  FE-side inputs (A, b: FEniCS's job in the reference) are resident in
HBM when the timed region starts whenever they fit (cfg2: A 13 GB; b always); cfg3's A (684 GB) is
produced per z-sub-slab inside the timed region and its time is reported separately.  Default
workload: cfg3 = 3D 256^3 p=3 (the configuration BASELINE.json's metric is quoted on).

N > 1: one process per GPU.  Launched by `python -m torch.distributed.run` (RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_* in the environment) the script is one rank; launched plainly with --gpus N it
spawns the N ranks itself.  The patch is split into z-slabs of dof planes; `n_gpus` and
`parallelism` are what the communicator reports, not what argv asked for.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable

WORKLOADS = {
    # name: (dim, degree, elements per direction)  -- BASELINE.json configs[1..4]
    "cfg2": (3, 2, 128),
    "cfg3": (3, 3, 256),
    "cfg1": (2, 2, 32),
    "cfg4": (2, 4, 256),      # demos/biharmonic: (-1,1)^2, two clamped layers, default solver = direct LU
    "cfg5": (2, 3, 128),      # NURBS quarter annulus, 3 fields, non-symmetric A on the 3-field pattern, GMRES
}
DEFAULT_SOLVER = {"cfg4": "lu", "cfg5": "gmres"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def counts(d, p, nel, nf=1):
    nnzM1 = 2 + (nel - 1) * p + nel * (p - 1) * (p + 1)
    nnzA1 = (nel - 1) * (2 * p + 1) + 2 * (p + 1) + nel * (p - 1) * (p + 1)
    nnzK1 = (nel + p) * (2 * p + 1) - p * (p + 1)
    return {"rows_fe": nf * (nel * p + 1) ** d, "ncp": nf * (nel + p) ** d, "nnzM": nf * nnzM1 ** d,
            "nnzA": nf * nf * nnzA1 ** d, "nnzK": nf * nf * nnzK1 ** d}


def ptap_bytes(cnt):
    # 12 nnz(A) + 2 * 12 nnz(M) + 12 nnz(K) (+ row pointers)
    return 12.0 * cnt["nnzA"] + 24.0 * cnt["nnzM"] + 12.0 * cnt["nnzK"] + 8.0 * (2 * cnt["rows_fe"] + cnt["ncp"])


def _elevate(Pw):
    """degree elevation by one of a Bezier curve given by homogeneous control points [n+1, c]"""
    n = Pw.shape[0] - 1
    out = np.zeros((n + 2, Pw.shape[1]))
    out[0], out[-1] = Pw[0], Pw[-1]
    for i in range(1, n + 1):
        a = i / float(n + 1)
        out[i] = a * Pw[i - 1] + (1.0 - a) * Pw[i]
    return out


def _refined_net(coarse, fine, Pw):
    """control net of the same curve on a refined knot vector: interpolation at the fine Greville points"""
    nf = fine.getNcp()
    us = np.array([fine.greville(r) for r in range(nf)])
    Nf, Nc = np.zeros((nf, nf)), np.zeros((nf, coarse.getNcp()))
    rows = np.arange(nf)[:, None]
    _, idx, val = fine.evalBatch(us)                     # (one device call per spline instead of one per point)
    Nf[rows, idx] = val
    _, idx, val = coarse.evalBatch(us)
    Nc[rows, idx] = val
    return np.linalg.solve(Nf, Nc @ Pw)


def quarter_annulus_mesh(p, nel):
    """cfg5's geometry: exact quarter annulus, radii 1..2, rational arc knot-refined to nel x nel
    elements of degree p -- deterministic, no RNG."""
    from tigar_amd.BSplines import BSpline1, uniformKnots
    from tigar_amd.NURBS import NURBSControlMesh
    w = 1.0 / np.sqrt(2.0)
    arc = np.array([[1.0, 0.0, 1.0], [w, w, w], [0.0, 1.0, 1.0]])
    rad = np.array([[1.0], [2.0]])
    while arc.shape[0] < p + 1:
        arc = _elevate(arc)
    while rad.shape[0] < p + 1:
        rad = _elevate(rad)
    rad = rad[:, 0]
    kv = uniformKnots(p, 0.0, 1.0, nel)
    coarse = BSpline1(p, [0.0] * (p + 1) + [1.0] * (p + 1))
    fine = BSpline1(p, kv)
    Pw = np.zeros((p + 1, p + 1, 3))
    for i in range(p + 1):
        Pw[i, :, 0], Pw[i, :, 1], Pw[i, :, 2] = rad[i] * arc[:, 0], rad[i] * arc[:, 1], arc[:, 2]
    Pr = np.stack([_refined_net(coarse, fine, Pw[:, j, :]) for j in range(p + 1)], axis=1)
    Pf = np.stack([_refined_net(coarse, fine, Pr[i, :, :]) for i in range(Pr.shape[0])], axis=0)
    return NURBSControlMesh([p, p], [kv, kv], Pf)


def rational_volume_mesh(p, nel):
    """The mapped-geometry companion's patch: a smooth rational volume map of degree p on nel^3 elements
    -- control points = a non-affine, non-separable map of the Greville points, weights varying; a control net in
    homogeneous coordinates as tIGAr/NURBS.py:46-74 reads it from igakit -- deterministic, no RNG."""
    from tigar_amd.BSplines import uniformKnots
    from tigar_amd.NURBS import NURBSControlMesh
    kv = np.asarray(uniformKnots(p, 0.0, 1.0, nel), dtype=np.float64)
    g = np.array([np.sum(kv[i + 1:i + p + 1]) / p for i in range(len(kv) - p - 1)])
    g0, g1, g2 = g[:, None, None], g[None, :, None], g[None, None, :]
    w = 1.0 + 0.25 * g0 * g1 + 0.1 * g2
    C = np.empty((len(g), len(g), len(g), 4))
    C[..., 0] = w * (g0 + 0.15 * g1 * g2)
    C[..., 1] = w * (g1 + 0.2 * g0 ** 2 - 0.1 * g2)
    C[..., 2] = w * (g2 * (1.0 + 0.3 * g0) + 0.05 * np.sin(2.0 * g1))
    C[..., 3] = w
    return NURBSControlMesh([p] * 3, [kv] * 3, C)


def hashed_values(n, seed):
    """counter-based hash -> doubles in (-1, 1): value k depends on (seed, k) only (splitmix64)"""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def spmv_bytes(nnzK, ncp):
    # fp64 values + int32 columns, row pointer, x read once, y written once
    return 12 * nnzK + 4 * (ncp + 1) + 16 * ncp


# ------------------------------------------------------------------------------------ the path
def run(args, wl, d, p, nel):
    from tigar_amd import device as dev
    from tigar_amd import common as tc
    from tigar_amd.common import (EqualOrderSpline, ExtractedSpline, PETScKrylovSolver, PETScLUSolver, Function,
                                  TensorFunctionSpace)
    from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
    from tigar_amd.forms import LaplaceForm, SeparableLoadForm, BiharmonicForm, SumOfSeparableLoads, NodalLoadForm
    mapped = getattr(args, "geometry", "identity") == "volume"
    assert not mapped or (d == 3 and wl in ("cfg2", "cfg3")), "the rational volume map is a 3-D Poisson configuration"

    comm = tc.worldcomm                       # size / rank from the launcher's environment
    rank, world = comm.rank, comm.size
    transport = comm.transport()
    dcomm = comm.device()                     # RCCL / IPC / host-staged communicator, None on one rank
    if rank == 0:
        log("[bench] device:", dev.device_info(), "ranks", world,
            "communicator", dcomm.info() if dcomm is not None else None)
    nf, nlayers, lo = 1, 1, 0.0
    method = args.solver if args.solver != "auto" else DEFAULT_SOLVER.get(wl, "cg")
    if wl == "cfg5":
        controlMesh, nf, nlayers = quarter_annulus_mesh(p, nel), 3, 2
    elif mapped:
        controlMesh = rational_volume_mesh(p, nel)
    else:
        if wl == "cfg4":
            lo, nlayers = -1.0, 2
        controlMesh = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, lo, 1.0, nel) for _ in range(d)])
    basis = controlMesh.getScalarSpline()
    f1 = lambda x: np.sin(np.pi * x)
    if wl == "cfg4":
        # demos/biharmonic/biharmonic.py:100-122: u = (cos(pi x)+1)(cos(pi y)+1)
        c, c1, pi4 = (lambda x: np.cos(np.pi * x)), (lambda x: np.cos(np.pi * x) + 1.0), np.pi ** 4
        lap = BiharmonicForm()
        load = SumOfSeparableLoads([([c, c1], pi4), ([c, c], 2 * pi4), ([c1, c], pi4)])
        exact1 = [c1, c1]
    else:
        lap = LaplaceForm()
        load = SeparableLoadForm([f1] * d, scale=d * np.pi ** 2)
        exact1 = [f1] * d

    # ---- FE-side inputs resident in HBM before the timed region, when they fit (not part of the path)
    cnt = counts(d, p, nel, nf)
    t0 = time.perf_counter()
    V_in = TensorFunctionSpace([basis.generateMesh(degree=p)], "Lagrange")
    free_b = dev.mem_info()[0] + dev.pool_stats()[0]
    a_resident = world == 1 and args.slab != 1 and not mapped and \
        12.0 * (2 * cnt["nnzM"] + cnt["nnzA"] + 2 * cnt["nnzK"]) <= 0.6 * free_b
    if wl == "cfg5":
        # a deterministic non-symmetric, diagonally dominant matrix on the 3-field Q_p pattern, values
        # from a counter-based hash (seed 0); uploaded from the host like a matrix dolfin assembled (no certificate)
        import scipy.sparse as sp
        assert world == 1, "cfg5 is a replicas-only configuration (SURVEY.md 8e)"
        pat = lap.assemble_matrix(V_in).to_scipy().tocsr()
        pat.sort_indices()
        blocks = [[None] * nf for _ in range(nf)]
        for a in range(nf):
            for b in range(nf):
                Bk = pat.copy()
                Bk.data = 0.05 * hashed_values(Bk.nnz, 3 * a + b)
                blocks[a][b] = Bk if a != b else (Bk + 4.0 * sp.identity(pat.shape[0], format="csr")).tocsr()
        A_host = sp.bmat(blocks, format="csr")
        assert A_host.nnz == cnt["nnzA"], (A_host.nnz, cnt["nnzA"])
        A_in = dev.DeviceCSR.from_scipy(A_host)
        b_in = dev.DeviceVector(data=hashed_values(A_host.shape[0], 1000))
        del A_host, blocks, pat
        a_resident = True
    else:
        A_in = lap.assemble_matrix(V_in) if a_resident else None
        b_in = load.assemble_vector(V_in) if (world == 1 and not mapped) else None
    dev.sync()
    t_input_pre = time.perf_counter() - t0
    if not a_resident:
        os.environ["TIGAR_IMPLICIT_M"] = "1"        # A arrives in row blocks: M stays implicit, K is built slab by slab
    if rank == 0:
        log("[bench] inputs resident before the timed region: A %s, b %s (%.3f s, untimed)"
            % ("yes" if a_resident else "no (row blocks produced inside the step)",
               "yes" if b_in is not None else "no (rank-local rows produced inside the step)", t_input_pre))

    stages = {}
    state = {}

    def make_solver():
        if method == "lu":
            return PETScLUSolver()
        solver = PETScKrylovSolver(method, args.pc)
        if args.pc == "chebyshev":
            solver.parameters["chebyshev_degree"] = args.cheb_degree
        solver.parameters["relative_tolerance"] = args.rtol
        solver.parameters["maximum_iterations"] = 100000
        return solver

    def step(record):
        ts = [time.perf_counter()]
        rec = {}

        def mark(name):
            dev.sync()
            ts.append(time.perf_counter())
            rec[name] = ts[-1] - ts[-2]

        gen = EqualOrderSpline(comm, nf, controlMesh)      # generateM_control / generateM (or implicit) / cpFuncs
        for field in range(nf):
            sp_ = gen.getScalarSpline(field)
            if wl == "cfg5":                               # shell-like clamp: two layers on one edge, every field
                gen.addZeroDofs(field, sp_.getSideDofs(0, 0, nLayers=nlayers))
                continue
            for direction in range(d):
                for side in (0, 1):
                    gen.addZeroDofs(field, sp_.getSideDofs(direction, side, nLayers=nlayers))
        mark("extract")
        spline = ExtractedSpline(gen, 2 * p)               # M^T
        mark("transpose")
        spline.stage_timers = {}
        # (mapped geometry: the forms integrate on the mapped patch, F = cpFuncs[i] / cpFuncs[nsd] of THIS generator --
        #  dolfin.assemble with the spline's measures, tIGAr/common.py:917-945, 1206-1220 -- in row blocks, inside the step)
        lap_s = LaplaceForm(geometry=gen) if mapped else lap
        load_s = NodalLoadForm(1.0, gen) if mapped else load
        K = spline.extractMatrix(A_in) if a_resident else spline.assembleMatrix(lap_s)   # M^T A M + zeroRowsColumns
        mark("ptap")
        rhs = spline.extractVector(b_in) if b_in is not None else spline.assembleVector(load_s)   # M^T b + BCs
        mark("mtb")
        solver = make_solver()
        spline.setSolverOptions(linearSolver=solver)
        u = Function(spline.V, spline.localFERange() if world > 1 else None)
        U = spline.solveLinearSystem(K, rhs, u)            # Krylov / LU + prolongation u = M U
        mark("solve")
        t_in = spline.stage_timers.get("input", 0.0)
        rec["fe_input"] = t_in
        rec["ptap"] -= t_in
        if record:
            for k, v in rec.items():
                stages.setdefault(k, []).append(v)
        state.update(gen=gen, spline=spline, K=K, U=U, u=u, solver=solver, rhs=rhs)
        if os.environ.get("TIGAR_TRACE"):
            pb, nb, nl = dev.pool_stats()
            fr, tot = dev.mem_info()
            log("[bench] pool: %.1f GB free in %d blocks, %d blocks live; device free %.1f GB; sub-slab timers %s"
                % (pb / 2 ** 30, nb, nl, fr / 2 ** 30, {k: round(v, 4) for k, v in spline.stage_timers.items()}))

    def barrier():
        dev.sync()
        transport.barrier()

    for _ in range(args.warmup):
        state.clear()        # (the previous step's K must not stay alive beside the one being assembled)
        step(False)
    dev.prof_reset()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        state.clear()
        step(True)
    barrier()
    elapsed = transport.allreduce_max(time.perf_counter() - t_start)

    gen, spline, K, u, solver = state["gen"], state["spline"], state["K"], state["u"], state["solver"]
    ncp = sum(gen.getNcp(f) for f in range(nf))
    nnzK_local, ncp_local = K.nnz, K.shape[0]
    nnzK = nnzK_local if dcomm is None else int(round(dcomm.allreduce_sum([float(nnzK_local)])[0]))
    spmv_ms, spmv_n = dev.prof_get(0)
    ksp_persistent = dev.prof_get(6)[1]
    ptap_certified = dev.prof_get(3)[1]                   # x passes that took A's pattern from its certificate
    comm_host_waits = dev.prof_get(4)[1]
    sell_classes, sell_padded = K.spmv_sell(True)         # which product kernel the solver used
    K.spmv_sell(False)
    symgrid_solves = dev.prof_get(7)[1]                   # CG solves on the half-storage copy (csrc/tg_symgrid.hip)
    symgrid = K.mult_symgrid(row0=(dcomm.g0 if dcomm is not None else 0))[1] if symgrid_solves > 0 else None
    its = solver.last.get("iterations", 0) if solver.last else 0
    mean_stages = {k: float(np.mean(v)) for k, v in stages.items()}

    # ---- per-rank self-check of the solve that was timed: || rhs - K U || over the rank's rows (halo through the
    # communicator), relative to || rhs ||; a communicator that lost or reordered an exchange shows up here
    self_check = None
    try:
        U_loc, rhs_loc = state["U"], state["rhs"]
        if dcomm is not None:
            xext = dcomm.halo_extend(U_loc)
            g0 = dcomm.g0 - dcomm.halo_lo
            KU = K.mult_offset(xext, g0)
        else:
            KU = K.mult(U_loc)
        if KU is not None:
            rr = rhs_loc.get_local() - KU.get_local()
            num, den = float(rr @ rr), float(rhs_loc.get_local() @ rhs_loc.get_local())
            if dcomm is not None:
                num, den = [float(v) for v in dcomm.allreduce_sum([num, den])]
            self_check = float(np.sqrt(num / den)) if den > 0 else 0.0
            if rank == 0:
                log("[bench] self-check ||rhs - K U|| / ||rhs|| over all ranks: %.3e" % self_check)
    except Exception as e:                                # (diagnostic only)
        log("[bench] self-check skipped:", repr(e))

    # ---- companion figures, measured live after the timed loop (one extra step each):
    #  * "materialised": the FE matrix written in row blocks inside the step and read back by the PtAP (what the step
    #    costs when the form is not a Kronecker sum the PtAP can fuse -- TIGAR_PTAP_FUSED=0);
    #  * "pattern_verified": additionally with A's pattern verified entry by entry (a matrix handed over without the
    #    library's certificate, e.g. uploaded from dolfin -- TIGAR_PTAP_VERIFY=1)
    companions = {}
    fused = (not a_resident) and mean_stages.get("fe_input", 1.0) < 1e-3 and ptap_certified == 0 and wl in ("cfg1", "cfg2", "cfg3")

    def extra_step(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            state.clear()
            barrier()
            t1 = time.perf_counter()
            step(False)
            barrier()
            return transport.allreduce_max(time.perf_counter() - t1)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    if args.companion and fused:
        extra_step({"TIGAR_PTAP_FUSED": "0"})       # (untimed: the allocator's pool has no blocks of the row-block sizes yet)
        companions["materialised"] = extra_step({"TIGAR_PTAP_FUSED": "0"})
        companions["pattern_verified"] = extra_step({"TIGAR_PTAP_FUSED": "0", "TIGAR_PTAP_VERIFY": "1"})
    elif args.companion and ptap_certified > 0:
        companions["pattern_verified"] = extra_step({"TIGAR_PTAP_VERIFY": "1"})
    if companions:
        gen, spline, K, u, solver = state["gen"], state["spline"], state["K"], state["u"], state["solver"]
    verified = companions.get("pattern_verified")

    nodal_error = None
    if args.check and rank == 0 and wl != "cfg5" and not mapped:
        # manufactured solution at the FE nodes this rank owns
        grid = spline.V.grids[0]
        r0, r1 = spline.localFERange()
        n0 = grid.shape()
        uh = u.vector().get_local()
        # every node for small problems, every 97th for large ones (cfg3: 4.7 M of 455 M nodes)
        idx = np.arange(r0, r1) if uh.size <= 40e6 else np.arange(r0, r1, 97)
        uh = uh if uh.size <= 40e6 else uh[idx - r0]
        exact = np.ones(idx.size)
        stride = 1
        for k in range(d):
            exact *= exact1[k](grid.axes[k][(idx // stride) % n0[k]])
            stride *= n0[k]
        nodal_error = float(np.max(np.abs(uh - exact)))
        log("[bench] max nodal error vs manufactured solution (rank 0 rows): %.3e" % nodal_error)
    if rank == 0:
        log("[bench] stages (mean s):", {k: round(v, 5) for k, v in mean_stages.items()}, "iterations:", its,
            "nnz(K) global:", nnzK, "M implicit:", bool(getattr(gen.M, "is_implicit", False)))
    # per-rank stage times (a first multi-GPU run has to be diagnosable from its one line: which rank, which stage)
    stage_names = sorted(mean_stages)
    per_rank = None
    if world > 1:
        tab = np.zeros((world, len(stage_names) + 3))
        tab[rank, :len(stage_names)] = [mean_stages[k] for k in stage_names]
        tab[rank, -3], tab[rank, -2] = float(K.shape[0]), float(K.nnz)
        # what a rank RECEIVES per product of the solve: its halo of the direction vector, p planes from either z-neighbour
        tab[rank, -1] = 8.0 * float(dcomm.halo_lo + dcomm.halo_hi) if dcomm is not None else 0.0
        transport.allreduce_sum(tab)
        per_rank = [dict({k: round(float(tab[r, i]), 6) for i, k in enumerate(stage_names)}, rank=r, dof_rows=int(tab[r, -3]),
                         nnz_K=int(tab[r, -2]), halo_bytes_per_product=int(tab[r, -1])) for r in range(world)]
    info = dcomm.info() if dcomm is not None else (0, 1, "none")
    ndev = dev.device_count()
    n_used = 1
    if info[1] > 1:
        devs = dcomm.rank_devices() if info[2] == "ipc" else None
        n_used = len(set(devs)) if devs and all(v is not None for v in devs) else min(info[1], ndev)
    return {"ncp": ncp, "nnzK": nnzK, "nnzK_local": nnzK_local, "ncp_local": ncp_local, "elapsed": elapsed,
            "spmv_ms_total": spmv_ms, "spmv_count": spmv_n, "ksp_persistent": ksp_persistent, "iterations": its, "stages": mean_stages,
            "t_input": mean_stages.get("fe_input", 0.0), "t_input_in_timed_region": not a_resident,
            "t_input_pre": t_input_pre, "sub_planes": getattr(spline._slab, "sub_planes_used", spline._slab.sub_planes) if spline._slab is not None else None,
            "sell_classes": sell_classes, "sell_padded": sell_padded, "symgrid": symgrid, "symgrid_solves": symgrid_solves, "implicit_M": bool(getattr(gen.M, "is_implicit", False)),
            "ptap_certified": int(ptap_certified), "nodal_error": nodal_error, "method": method, "nf": nf,
            "solver_last": {k: v for k, v in (solver.last or {}).items() if isinstance(v, (int, float, str, bool))},
            "self_check": self_check, "verified_step_s": verified, "materialised_step_s": companions.get("materialised"),
            "fused": bool(fused), "comm_host_waits": int(comm_host_waits),
            "comm_world": info[1], "comm_kind": info[2], "n_devices_used": n_used, "per_rank_stages": per_rank,
            "comm_requested": getattr(dcomm, "requested_kind", None) if dcomm is not None else None,
            "comm_fallback": getattr(dcomm, "fallback_notes", None) if dcomm is not None else None,
            "comm_devices": (dcomm.rank_devices() if (dcomm is not None and info[2] == "ipc") else None),
            "devices_visible": ndev, "mapped": mapped}


# ------------------------------------------------------------------------------------ CPU baseline
def _cpu_sample(d, p, nel, threads, factored=False):
    """One pass of the path on the host for a d-D degree-p patch with nel^d elements: the oracle's C + OpenMP
    restatement (oracle/tigar_oracle_c.c: the CSR algorithms PETSc AIJ runs on the CPU -- row-wise generateM,
    Gustavson PtAP + MatZeroRowsColumns, scatter-add M^T b, Jacobi-CG with PETSc's convergence test) on `threads`
    threads.  ``factored``: M^T A M as the direction-by-direction product P_z^T(P_y^T(P_x^T A P_x)P_y)P_z built from the
    same C Gustavson kernel -- the algorithm the GPU path uses, on the CPU, so that the algorithmic and the hardware gain
    separate.  FE inputs A, b are generated beforehand (untimed, as on the GPU side)."""
    import scipy.sparse as sp
    from oracle import tigar_oracle as O
    from oracle import tigar_oracle_c as OC
    OC.set_threads(threads)
    t0 = time.perf_counter()
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    f = lambda x: np.sin(np.pi * x)
    # FE inputs from the device generator (identical matrices, seconds instead of half a minute of scipy.kron)
    from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
    from tigar_amd.common import TensorFunctionSpace
    from tigar_amd.forms import LaplaceForm, SeparableLoadForm
    basis = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0.0, 1.0, nel) for _ in range(d)]).getScalarSpline()
    V_in = TensorFunctionSpace([basis.generateMesh(degree=p)], "Lagrange")
    A = LaplaceForm().assemble_matrix(V_in).to_scipy()
    b = SeparableLoadForm([f] * d, scale=d * np.pi ** 2).assemble_vector(V_in).get_local()
    t_in = time.perf_counter() - t0
    t0 = time.perf_counter()
    M = OC.generate_M_tensor(s)
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    t1 = time.perf_counter()
    # (the intermediate A*M is held in blocks of at most 3e8 entries = 3.6 GB: tgo_ptap_blocked)
    if factored:
        M1 = [O.generate_M_tensor(O.BSpline([p], [O.uniform_knots(p, 0., 1., nel)])).tocsr() for _ in range(d)]
        K = OC.ptap_sum_factorised(M1, A, zd, 1.0, max_am_entries=int(3e8))
    else:
        K = OC.extract_matrix(M, A, zd, max_am_entries=int(3e8))
    t2 = time.perf_counter()
    rhs = OC.extract_vector(M, b, zd)
    t3 = time.perf_counter()
    U, its, _ = OC.cg_jacobi(K, rhs, rtol=1e-6)
    u = OC.spmv(M, U)
    t4 = time.perf_counter()
    ncp = s.getNcp()
    X, _ = O.fe_node_grid(s)
    err = float(np.max(np.abs(u - np.prod(np.sin(np.pi * X), axis=1))))
    return {"nel": nel, "ncp": ncp, "threads": OC.num_threads(), "extract_s": t1 - t0, "ptap_s": t2 - t1, "mtb_s": t3 - t2,
            "solve_s": t4 - t3, "its": its, "total_s": t4 - t0, "err": err, "inputs_s": t_in, "nnzK": int(K.nnz)}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(d, p, nel_all, nel_one, nel_target, its_target, gpu_general=None):
    """the CPU restatement on ONE thread and on all cores the box grants, on bounded samples of
    the same workload (same d, p; fewer elements), plus the stated extrapolation to the benchmark size (set-up linear
    in the DoFs, solve = iterations x per-iteration time) and the sum-factorised product on one thread, so that the
    algorithmic gain is visible apart from the hardware gain.  A reported baseline, not a target."""
    from oracle import tigar_oracle_c as OC
    cores = min(32, OC.usable_cores())       # (the samples are small; more threads only add barrier cost)
    t0 = time.perf_counter()
    allc = _cpu_sample(d, p, nel_all, cores)
    if time.perf_counter() - t0 > 60.0 and d == 3:
        nel_one = max(4, min(nel_one, 24))       # (a slow host: keep the whole baseline within a few minutes)
    one = _cpu_sample(d, p, nel_one, 1)
    fac = _cpu_sample(d, p, nel_one, 1, factored=True)
    log("[bench] cpu baseline: %.1f s in all" % (time.perf_counter() - t0))

    def extrapolate(smp):
        ncp_t = (nel_target + p) ** d
        setup = (smp["extract_s"] + smp["ptap_s"] + smp["mtb_s"]) * ncp_t / smp["ncp"]
        solve = smp["solve_s"] / max(smp["its"], 1) * its_target * ncp_t / smp["ncp"]
        return {"seconds": setup + solve, "DoF_per_s": ncp_t / (setup + solve)}
    fmt = lambda q: ("%dD p=%d %d^%d elements (%d DoFs) on %d thread%s: extract %.2fs, M^T A M %.2fs, M^T b %.2fs, "
                     "CG(%d its)+prolongation %.2fs; max nodal error %.1e"
                     % (d, p, q["nel"], d, q["ncp"], q["threads"], "s" if q["threads"] > 1 else "", q["extract_s"], q["ptap_s"],
                        q["mtb_s"], q["its"], q["solve_s"], q["err"]))
    like = None
    if gpu_general:
        # like against like: both sides Gustavson-class products of arbitrary sparse operands (nothing assumed about A
        # or M) -- not the fused tensor-pattern path against Gustavson
        cpu_ext = extrapolate(allc)["DoF_per_s"]
        like = dict(gpu_general, cpu_all_cores_extrapolated_DoF_per_s=cpu_ext, ratio=gpu_general["value"] / cpu_ext)
    return {"value": allc["ncp"] / allc["total_s"], "unit": "DoF/s", "cores": allc["threads"], "kind": "port",
            "cpu_model": _cpu_model(), "nproc": os.cpu_count(), "gpu_general_path_over_cpu_gustavson": like,
            "sample": fmt(allc) + "; C+OpenMP restatement of the PETSc AIJ algorithms (oracle/tigar_oracle_c.c); inputs "
                      "%.2fs untimed" % allc["inputs_s"],
            "one_thread": {"value": one["ncp"] / one["total_s"], "unit": "DoF/s", "sample": fmt(one)},
            "one_thread_sum_factorised_ptap": {"ptap_s": fac["ptap_s"], "gustavson_ptap_s": one["ptap_s"],
                                               "value": fac["ncp"] / fac["total_s"], "unit": "DoF/s",
                                               "note": "M^T A M as P_z^T(P_y^T(P_x^T A P_x)P_y)P_z, every stage by the same C "
                                                       "Gustavson kernel (the GPU path's algorithm on one CPU thread), same sample"},

            "extrapolated_to_benchmark_size": {
                "nel": nel_target, "cg_iterations": its_target,
                "all_cores": extrapolate(allc), "one_thread": extrapolate(one),
                "method": "set-up (extract + M^T A M + M^T b) linear in the DoFs; solve = per-iteration time of the sample "
                          "x DoF ratio x the iteration count measured on the GPU at the benchmark size; the benchmark size "
                          "itself cannot be held on the host (M 271 GB, A 684 GB)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--nel", type=int, default=0, help="override elements per direction")
    ap.add_argument("--p", type=int, default=0)
    ap.add_argument("--d", type=int, default=0)
    ap.add_argument("--rtol", type=float, default=1e-6)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nel", type=int, default=0)
    ap.add_argument("--slab", type=int, default=-1, help="1: force the implicit-M / z-slab streaming path on one GPU")
    ap.add_argument("--solver", default="auto", help="cg | gmres | bicgstab | lu (auto: cg; cfg4 lu as demos/biharmonic; cfg5 gmres)")
    ap.add_argument("--pc", default="jacobi", help="jacobi (the benchmark's) | none | chebyshev (companion runs; cg only)")
    ap.add_argument("--cheb-degree", type=int, default=8)
    ap.add_argument("--live-traffic", type=int, default=-1,
                    help="1 / 0: measure the HBM traffic of the product kernel with rocprofv3 --pmc in a child run of this "
                         "command (default: when N = 1, the headline workload, rocprofv3 present and no profiler around this run)")
    ap.add_argument("--geometry", default="identity",
                    help="identity (the benchmark's unit cube) | volume: a smooth rational volume map -- the forms integrate on "
                         "the mapped patch (sum-factorised element matrices in row blocks, csrc/tg_assemble.hip)")
    ap.add_argument("--mapped-companion", type=int, default=-1,
                    help="1 / 0: after the timed run, one step of the same size on the rational volume map (default: on "
                         "for the headline workload on one rank)")
    ap.add_argument("--companion", type=int, default=1,
                    help="1: after the timed loop run one more step with the FE matrix' pattern verified entry by entry")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched plainly: become the launcher of N ranks (one per GPU), like torch.distributed.run would
        from tigar_amd.launch import spawn_local
        sys.exit(spawn_local(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        log("[bench] note: --gpus %d but the launcher started %d ranks; reporting what the communicator spans"
            % (args.gpus, world))
    wl = args.workload
    if wl == "auto":
        wl = "cfg3"        # the configuration BASELINE.json's metric is quoted on (256^3, p=3)
    d, p, nel = WORKLOADS[wl]
    if args.nel:
        nel = args.nel
    if args.p:
        p = args.p
    if args.d:
        d = args.d
    # HBM traffic of the product kernel, LIVE: two child runs of this very command (one step after one warm-up) under
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domains besides --kernel-trace, reads x2
    # on gfx950: tools/pmc_hbm.py) BEFORE this process touches the GPU (afterwards its own 150 GB leave a child no room).
    # Bounded (300 s), optional, never allowed to break the line; evaluated after the timed run.
    pmc_rows = None
    live = args.live_traffic
    if live < 0:
        import shutil
        profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        live = int(wl == "cfg3" and world == 1 and args.gpus == 1 and not args.nel and not args.p and not args.d and
                   shutil.which("rocprofv3") is not None and not profiled)
    if live and world == 1 and args.gpus == 1:
        import subprocess, tempfile
        dst = os.path.join(tempfile.mkdtemp(prefix="tigar_pmc_", dir="/tmp"), "pmc.json")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_hbm.py"), dst, "--", sys.executable, os.path.abspath(__file__),
               "--workload", wl, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--companion", "0", "--live-traffic", "0",
               "--rtol", repr(args.rtol), "--solver", args.solver, "--check", "0"]
        try:
            env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_"))}
            subprocess.run(cmd, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, check=False)
            pmc_rows = json.load(open(dst))["kernels"]
        except Exception as e:                          # noqa: BLE001
            log("[bench] live PMC pass skipped: %s" % (e,))
    res = run(args, wl, d, p, nel)
    # ---- companion: the same size on a MAPPED patch (rational volume), measured live: one warm-up and one step
    mapped_res = None
    mc = args.mapped_companion
    if mc < 0:
        mc = int(wl == "cfg3" and world == 1 and args.geometry == "identity" and bool(args.companion))
    if mc and args.geometry == "identity" and d == 3 and wl in ("cfg2", "cfg3"):
        import copy
        a2 = copy.copy(args)
        a2.geometry, a2.steps, a2.warmup, a2.companion, a2.check = "volume", 1, 1, 0, 0
        try:
            # the identity run's blocks (sliced K, walk buffers) sit idle in the caching allocator with sizes the mapped
            # run does not ask for; hand them back first, or every mapped step trims under pressure inside its stages
            import gc
            from tigar_amd import _lib
            gc.collect()
            _lib.lib().tg_pool_trim()
            mapped_res = run(a2, wl, d, p, nel)
        except Exception as e:                           # noqa: BLE001  (a companion must not break the line)
            log("[bench] mapped-geometry companion failed: %r" % (e,))
    if rank != 0:
        return
    cnt = counts(d, p, nel, res["nf"])

    n_gpus = res["n_devices_used"]
    ms_per_step = 1e3 * res["elapsed"] / args.steps
    value = res["ncp"] / (res["elapsed"] / args.steps)
    spmv_avg_s = (res["spmv_ms_total"] / max(res["spmv_count"], 1)) * 1e-3
    ncp_l, nnzK_l = res["ncp_local"], res["nnzK_local"]
    csr_bytes = spmv_bytes(nnzK_l, ncp_l)
    sell = res.get("sell_padded", 0) > 0 and os.environ.get("TIGAR_SPMV_SELL", "1") != "0"
    kernel = "k_spmv_sell" if sell else "k_spmv_lane"
    persistent = res.get("ksp_persistent", 0) > 0
    if persistent:
        # small systems: the whole CG loop is one kernel with K in registers (csrc/tg_krylov_small.hip); the "launch" below
        # is one ITERATION of it (product, inner products, two device-wide barriers, updates), not a product kernel
        kernel = "%s (per iteration; K register-resident)" % ("k_gmres_persistent" if res["method"] == "gmres" else "k_cg_persistent")
    # bytes the product kernel has to move in ITS format: the sliced copy streams 8 B per stored position
    # (values only; the offset dictionary is cache resident), class id + address per slice, x once and y once;
    # the general CSR kernel moves SURVEY.md section 8(d)'s CSR bytes
    fmt_bytes = (8.0 * res["sell_padded"] + 12.0 * ((ncp_l + 63) // 64) + 16.0 * ncp_l) if sell else float(csr_bytes)
    sym = res.get("symgrid") if (res.get("symgrid_solves", 0) > 0 and not persistent) else None
    if sym:
        # CG on a symmetric box-stencil K: the diagonal and the upper triangle only (16 B per pair of positions and row),
        # the LDS windows written to the staging array by the product kernel and read by the one that sums them, x, y
        kernel = "k_symgrid_spmv + k_symgrid_combine"
        fmt_bytes = float(sym["value_bytes"] + 2 * sym["staging_bytes"] + 16 * ncp_l)
    achieved = fmt_bytes / spmv_avg_s / 1e9 if spmv_avg_s > 0 else 0.0
    # HBM traffic of this kernel from the PMC passes (rocprofv3 --pmc in separate runs, as the microarchitecture
    # guide prescribes; committed under profiles/): an OFFLINE measurement, quoted only for the workload and
    # rank count it was taken on
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "r3_spmv_pmc_summary.json")
    if wl == "cfg3" and res["comm_world"] == 1 and not args.nel and not args.p and os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            if kernel in pmc["kernel"]:
                traffic = pmc["hbm_bytes_per_launch"]
                traffic_src = ("offline: profiles/r3_spmv_pmc_summary.json (rocprofv3 --pmc, FETCH_SIZE x2 gfx950 "
                               "correction + WRITE_SIZE, separate passes; not measured in this run)")
        except Exception:
            traffic = None
    if pmc_rows:                                     # the live passes taken before the run (see above)
        names = [k.strip() for k in kernel.split("+")]
        rows = [r for r in pmc_rows if any(nm in r["kernel"] for nm in names)]
        solves = 2                                      # (the child: one warm-up + one step)
        gated = 2 * solves if res["method"] == "cg" else 0     # products enqueued past convergence return at once
        if rows and rows[0]["launches"] > gated:
            n_real = rows[0]["launches"] - gated
            live_traffic = sum(r["hbm_read_GB_total_corrected"] + r["hbm_write_GB_total"] for r in rows) * 1e9 / n_real
            if 0.5 * fmt_bytes < live_traffic < 4.0 * fmt_bytes:      # (a sane count: else the offline figure stays)
                traffic = live_traffic
                traffic_src = ("live: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE in two child runs of this command "
                               "before the timed run (tools/pmc_hbm.py), %d launches that moved data" % n_real)
    par = "z-slab x%d ranks on %d GPU%s (%s)" % (res["comm_world"], n_gpus, "s" if n_gpus > 1 else "", res["comm_kind"]) \
        if res["comm_world"] > 1 else "1 GPU"
    step_s = res["elapsed"] / args.steps
    t_in_timed = res["t_input"] if res["t_input_in_timed_region"] else 0.0
    desc = {"cfg4": "B-spline biharmonic on (-1,1)^2, two clamped layers (demos/biharmonic)",
            "cfg5": "NURBS quarter annulus, 3 fields, hashed non-symmetric A on the 3-field pattern"}.get(
                wl, "Poisson on a rational volume map (mapped forms)" if res.get("mapped") else "B-spline Poisson")
    pc_name = {"jacobi": "Jacobi", "none": "unpreconditioned", "chebyshev": "Chebyshev(%d)-Jacobi" % args.cheb_degree}[args.pc]
    solver_desc = {"cg": "%s-CG rtol %.0e" % (pc_name, args.rtol), "gmres": "%s-GMRES(30) rtol %.0e" % (pc_name, args.rtol),
                   "bicgstab": "%s-BiCGStab rtol %.0e" % (pc_name, args.rtol),
                   "lu": "direct banded solve (the reference's default solver is a direct LU): " +
                         ("blocked Cholesky on the matrix cores, K found symmetric positive definite (csrc/tg_chol.hip)"
                          if (res.get("solver_last") or {}).get("factorisation") == "cholesky" else "LU with partial pivoting")
                   }[res["method"]]
    solver_api = {"cg": "PETScKrylovSolver('cg','%s')" % args.pc, "gmres": "PETScKrylovSolver('gmres','%s')" % args.pc,
                  "bicgstab": "PETScKrylovSolver('bicgstab','%s')" % args.pc, "lu": "PETScLUSolver()"}[res["method"]]
    out = {
        "metric": "DoF/s (extraction + M^T A M + M^T b + CG solve + prolongation)",
        "value": value, "unit": "DoF/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dD %d^%d elements p=%d %s, Q_p extraction, %s" % (wl, d, nel, d, p, desc, solver_desc),
                   "api": "EqualOrderSpline -> ExtractedSpline.assembleMatrix/extractMatrix, assembleVector/"
                          "extractVector, solveLinearSystem(%s)" % solver_api,
                   "dofs": res["ncp"], "fe_rows": cnt["rows_fe"], "nnz_M": cnt["nnzM"], "nnz_A": cnt["nnzA"],
                   "nnz_K": res["nnzK"], "cg_iterations": res["iterations"], "solver": res["solver_last"],
                   "M_implicit": res["implicit_M"],
                   "max_nodal_error_vs_manufactured_solution": res.get("nodal_error"),
                   "fe_matrix_pattern": ("never materialised: the form is a Kronecker sum of 1-D matrices and the first PtAP pass "
                                         "forms the entries itself, bit for bit as tg_kron_sum_csr would have written them "
                                         "(SURVEY 8d: fused A-generation; the PtAP roofline below is quoted on the "
                                         "MATERIALISED byte count)" if res.get("fused") else
                                         "certified by the assembly kernel that wrote it (tg_kron_sum_csr): the PtAP does not "
                                         "re-read the column indices; TIGAR_PTAP_VERIFY=1 verifies them entry by entry"
                                         if res.get("ptap_certified", 0) > 0
                                         else "verified entry by entry while the PtAP reads it"),
                   "stages_s": {k: round(v, 6) for k, v in res["stages"].items()},
                   "fe_input_generation_s": round(res["t_input"], 6),
                   "fe_input_inside_timed_region": bool(res["t_input_in_timed_region"]),
                   # SURVEY.md 8(d) excludes the FE assembly of A, b from the metric; `value` keeps it when it has to
                   # happen inside the step (conservative), this is the figure by the survey's definition
                   "value_excluding_fe_input": res["ncp"] / max(1e-12, step_s - t_in_timed),
                   # the same step with A's pattern verified entry by entry (a matrix handed over without certificate,
                   # e.g. uploaded from dolfin): one extra step measured live after the timed loop
                   "value_pattern_verified": (res["ncp"] / res["verified_step_s"]) if res.get("verified_step_s") else None,
                   "ms_per_step_pattern_verified": 1e3 * res["verified_step_s"] if res.get("verified_step_s") else None,
                   # the step with the FE matrix written in row blocks and read back (no fusion of a Kronecker-sum form
                   # into the first PtAP pass): what `value` was before round 3
                   "value_fe_matrix_materialised": (res["ncp"] / res["materialised_step_s"]) if res.get("materialised_step_s") else None,
                   "ms_per_step_fe_matrix_materialised": 1e3 * res["materialised_step_s"] if res.get("materialised_step_s") else None,
                   "fe_matrix_fused_into_ptap": res.get("fused"),
                   # the same size on a mapped patch (rational volume map; the forms integrate with the spline's measures,
                   # element matrices sum-factorised in row blocks inside the step): one step measured live after the timed run
                   "value_mapped_geometry": (mapped_res["ncp"] / mapped_res["elapsed"]) if mapped_res else None,
                   "mapped_geometry": ({"ms_per_step": 1e3 * mapped_res["elapsed"],
                                        "stages_s": {k: round(v, 6) for k, v in mapped_res["stages"].items()},
                                        "cg_iterations": mapped_res["iterations"],
                                        "self_check_rel_residual": mapped_res.get("self_check"),
                                        "ptap_certified_passes": mapped_res.get("ptap_certified"),
                                        "geometry": "rational volume map: control points = smooth non-affine map of the Greville "
                                                    "points, varying weights (bench.rational_volume_mesh); f = 1, zero Dirichlet "
                                                    "data on all faces",
                                        "forms": "LaplaceForm(geometry=gen), NodalLoadForm(1.0, gen)"} if mapped_res else None),
                   "self_check_rel_residual_all_ranks": res.get("self_check"),
                   "communicator_host_waits_in_timed_steps": res.get("comm_host_waits"),
                   "ptap": {"stage_s": res["stages"].get("ptap"), "algorithmic_bytes": ptap_bytes(cnt),
                            "achieved_GBps": ptap_bytes(cnt) / max(1e-12, res["stages"].get("ptap", 0.0)) / 1e9,
                            "frac_of_hbm_peak": ptap_bytes(cnt) / max(1e-12, res["stages"].get("ptap", 0.0)) / 1e9 / HBM_PEAK_GBS,
                            "bytes_definition": "SURVEY.md 8d: 12 nnz(A) + 24 nnz(M) + 12 nnz(K) + row pointers, stage wall time"},
                   "sub_planes": res.get("sub_planes"),
                   "ranks": res["comm_world"], "communicator": res["comm_kind"],
                   "communicator_requested": res.get("comm_requested"),
                   "communicator_fallback": res.get("comm_fallback") or None,
                   "devices_visible": res.get("devices_visible"), "device_of_rank": res.get("comm_devices"),
                   "per_rank_stages_s": res.get("per_rank_stages"),
                   "parallelism": par},
        "roofline": {"bound": "hbm", "kernel": "%s (K u in %s)" % (kernel, res["method"].upper()), "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "launches": res["spmv_count"], "avg_launch_ms": spmv_avg_s * 1e3,
                     "bytes_per_launch": fmt_bytes,
                     "bytes_definition": ("sliced, pattern-compressed copy of K's values: 8 B per stored position + "
                                          "12 B per 64-row slice + x read once + y written once (what this kernel "
                                          "must move; no column indices are streamed)") if sell and not sym else
                                         ("half storage of a symmetric box-stencil K (csrc/tg_symgrid.hip): 16 B per pair of "
                                          "stored positions and row (the diagonal and the entries above it, each used for "
                                          "its row and for the transposed entry) + the LDS windows written to and read from "
                                          "the staging array + x read once + y written once; the time is that of both "
                                          "kernels of a product") if sym else
                                         "CSR: 12 B per entry + row pointers + x + y (SURVEY.md 8d)",
                     "csr_bytes_per_launch": csr_bytes,
                     "effective_csr_GBps": csr_bytes / spmv_avg_s / 1e9 if spmv_avg_s > 0 else 0.0,
                     "effective_csr_frac_of_peak": csr_bytes / spmv_avg_s / 1e9 / HBM_PEAK_GBS if spmv_avg_s > 0 else 0.0},
    }
    if persistent:
        out["roofline"]["bytes_definition"] = ("SURVEY.md 8d's CSR bytes of one product K u, for reference only: the kernel holds K in "
                                               "registers and reads it ONCE per solve; per iteration it moves the vector u and the "
                                               "partial sums -- the iteration is bound by two device-wide barriers, not by HBM")
        out["roofline"]["bytes_per_launch"] = float(csr_bytes)
        out["roofline"]["achieved"] = csr_bytes / spmv_avg_s / 1e9 if spmv_avg_s > 0 else 0.0
        out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    if res["spmv_count"] == 0:
        # no Krylov product was timed (direct solve): the dominant kernel of the path is the triple product
        pt = out["config"]["ptap"]
        out["roofline"] = {"bound": "hbm", "kernel": "M^T A M stage (all its kernels; wall time of the stage)",
                           "achieved": pt["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": pt["frac_of_hbm_peak"], "traffic": None,
                           "bytes_per_launch": pt["algorithmic_bytes"], "bytes_definition": pt["bytes_definition"]}
    if wl == "cfg3" and res["comm_world"] == 1 and not args.nel and not args.p:
        # the general-CSR contract keeps a tracked number: the same step with the fast path switched off (offline runs,
        # committed under profiles/; A arbitrary sparse in both, M Kronecker in the first, nothing assumed in the second)
        ref = {}
        for key, fn in (("fe_matrix_materialised_in_row_blocks", "r3_bench_cfg3_fe_matrix_materialised.json"),
                        ("fe_matrix_pattern_verified_entry_by_entry", "r3_bench_cfg3_pattern_verified.json"),
                        ("arbitrary_A_kronecker_M_line_kernels", "r4_bench_cfg3_general_line.json"),
                        ("fully_general_hash_ptap_M_slabs_materialised", "r4_bench_cfg3_general_hash.json"),
                        # round 6: the same contract -- M materialised chunk by chunk as a general CSR matrix, A in row blocks,
                        # nothing assumed about either -- through the element chunks (csrc/tg_elemsplit.hip)
                        ("fully_general_element_chunks_M_materialised", "r6_bench_cfg3_general_elements.json")):
            try:
                g = json.load(open(os.path.join(ROOT, "profiles", fn)))
                ref[key] = {"value": g["value"], "ms_per_step": g["ms_per_step"], "ptap_s": g["config"]["stages_s"]["ptap"],
                            "source": "offline: profiles/" + fn}
            except Exception:
                pass
        if ref:
            out["config"]["general_path_reference"] = ref
    if not args.no_cpu_baseline and res["comm_world"] == 1 and wl in ("cfg1", "cfg2", "cfg3"):   # (rank 0 at N=1 only)
        # bounded samples (SURVEY 8d / BASELINE.md section 4 plan 64^3 at p=3): all cores at 64^3 elements when the host has
        # the memory for it (A 10.7 GB + M, M^T 8.4 GB + blocks of A*M of 3.6 GB; checked against MemAvailable -- a
        # box must not be driven out of memory), else 48^3 / 40^3; one thread at half that edge (a one-thread 64^3
        # Gustavson product alone would take over a minute)
        avail = 0.0
        try:
            for line in open("/proc/meminfo"):
                if line.startswith("MemAvailable:"):
                    avail = float(line.split()[1]) * 1024.0
        except OSError:
            pass
        if d == 3:
            def need(n):      # bytes: A, M, M^T, scipy copies and the bounded intermediate, generously
                c = counts(d, p, n)
                return 2.2 * 12.0 * c["nnzA"] + 3.0 * 12.0 * c["nnzM"] + 8e9
            ladder = {2: (128, 96, 80), 3: (64, 48, 40), 4: (32, 24, 16)}.get(p, (16,))
            cpu_nel = args.cpu_nel or next((n for n in ladder if need(n) <= 0.6 * avail), ladder[-1])
            # one thread at three quarters of that edge (64 -> 48: BASELINE.md section 4's plan; the Gustavson intermediate
            # is held in blocks there as well; ~35 s)
            one_nel = max(4, int(round(cpu_nel * 0.75)))
        else:
            cpu_nel = args.cpu_nel or min(nel, 256)
            one_nel = max(4, cpu_nel // 2)
        gpu_general = None
        gref = out["config"].get("general_path_reference", {})
        for key in ("fully_general_hash_ptap_M_slabs_materialised", "fully_general_element_chunks_M_materialised"):   # (the last one found)
            if key in gref:
                gpu_general = {"value": gref[key]["value"], "unit": "DoF/s", "what": key, "source": gref[key]["source"]}
        out["cpu_baseline"] = cpu_baseline(d, p, cpu_nel, one_nel, nel, res["iterations"], gpu_general)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
