#!/usr/bin/env python3
"""
bench.py -- tIGAr extraction hot path on MI355X: extraction-operator build (generateM) ->
M^T A M + M^T b (extractMatrix / extractVector) -> Krylov solve (solveLinearSystem), on a
synthetic tensor-product B-spline Poisson patch.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|auto]

One "step" = one full pass of the hot path over the patch.  FE-side inputs (A, b: FEniCS's
job in the reference) are generated on the device BEFORE the timed region and are resident in
HBM when it starts whenever they fit (cfg2); for cfg3 (A = 684 GB) they are regenerated per
z-sub-slab inside the timed region and their time is reported separately.  Default workload:
cfg3 = 3D 256^3 p=3 (the configuration BASELINE.json's metric is quoted on), streamed through
one GPU in z-slabs or sharded over N GPUs.  Rank 0 prints ONE JSON line.
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
    # name: (dim, degree, elements per direction)  -- BASELINE.json configs[1], configs[2]
    "cfg2": (3, 2, 128),
    "cfg3": (3, 3, 256),
    "cfg1": (2, 2, 32),
    "cfg4": (2, 4, 256),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def counts(d, p, nel):
    nnzM1 = 2 + (nel - 1) * p + nel * (p - 1) * (p + 1)
    nnzA1 = (nel - 1) * (2 * p + 1) + 2 * (p + 1) + nel * (p - 1) * (p + 1)
    nnzK1 = (nel + p) * (2 * p + 1) - p * (p + 1)
    return {"rows_fe": (nel * p + 1) ** d, "ncp": (nel + p) ** d, "nnzM": nnzM1 ** d, "nnzA": nnzA1 ** d,
            "nnzK": nnzK1 ** d}


def spmv_bytes(nnzK, ncp):
    # fp64 values + int32 columns, row pointer, x read once, y written once
    return 12 * nnzK + 4 * (ncp + 1) + 16 * ncp


# ------------------------------------------------------------------------------------ 1 GPU
def run_single(args, d, p, nel):
    from tigar_amd import device as dev
    from tigar_amd.common import (EqualOrderSpline, ExtractedSpline, PETScKrylovSolver, Function,
                                  TensorFunctionSpace)
    from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
    from tigar_amd.forms import LaplaceForm, SeparableLoadForm

    info = dev.device_info()
    log("[bench] device:", info)
    kvecs = [uniformKnots(p, 0.0, 1.0, nel) for _ in range(d)]
    controlMesh = ExplicitBSplineControlMesh([p] * d, kvecs)
    basis = controlMesh.getScalarSpline()

    # ---- FE-side inputs, resident in HBM before the timed region (not part of the path)
    t0 = time.perf_counter()
    V_in = TensorFunctionSpace([basis.generateMesh(degree=p)], "Lagrange")
    A = LaplaceForm().assemble_matrix(V_in)
    f1 = lambda x: np.sin(np.pi * x)
    b = SeparableLoadForm([f1] * d, scale=d * np.pi ** 2).assemble_vector(V_in)
    dev.sync()
    t_input = time.perf_counter() - t0
    log("[bench] inputs: A %s nnz %d, b %d  (%.3f s, untimed)" % (A.shape, A.nnz, b.size(), t_input))

    stages = {}

    def step(record):
        ts = [time.perf_counter()]

        def mark(name):
            dev.sync()
            ts.append(time.perf_counter())
            if record:
                stages.setdefault(name, []).append(ts[-1] - ts[-2])

        gen = EqualOrderSpline(1, controlMesh)             # generateM_control / generateM / cpFuncs
        sp_ = gen.getScalarSpline(0)
        for direction in range(d):
            for side in (0, 1):
                gen.addZeroDofs(0, sp_.getSideDofs(direction, side))
        mark("extract")
        spline = ExtractedSpline(gen, 2 * p)               # explicit M^T
        mark("transpose")
        K = spline.extractMatrix(A)                        # M^T A M + zeroRowsColumns
        mark("ptap")
        rhs = spline.extractVector(b)                      # M^T b + BCs
        mark("mtb")
        solver = PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = args.rtol
        spline.setSolverOptions(linearSolver=solver)
        u = Function(spline.V)
        U = spline.solveLinearSystem(K, rhs, u)            # CG + prolongation u = M U
        mark("solve")
        return gen, spline, K, U, u, solver

    for _ in range(args.warmup):
        out = step(False)
        del out
    dev.prof_reset()
    dev.sync()
    t_start = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = None
        last = step(True)
    dev.sync()
    elapsed = time.perf_counter() - t_start
    gen, spline, K, U, u, solver = last
    ncp = K.shape[0]
    nnzK = K.nnz
    spmv_ms, spmv_n = dev.prof_get(0)
    Kd = K if hasattr(K, "spmv_sell") else None
    sell_classes, sell_padded = (0, 0)
    if Kd is not None:
        sell_classes, sell_padded = Kd.spmv_sell(True)
        Kd.spmv_sell(False)
    its = solver.last["iterations"]
    log("[bench] stages (mean s):", {k: round(float(np.mean(v)), 5) for k, v in stages.items()},
        "CG iterations:", its, "nnz(K):", nnzK, "nnz(M):", gen.M.nnz)

    # sanity: manufactured solution u = prod sin(pi x_k) at the FE nodes
    if args.check:
        X = spline.V.grids[0].coordinates() if ncp < 3e6 else None
        if X is not None:
            uh = u.vector().get_local()
            exact = np.prod(np.sin(np.pi * X), axis=1)
            log("[bench] max nodal error vs manufactured solution: %.3e" % np.max(np.abs(uh - exact)))

    result = {"ncp": ncp, "nnzK": nnzK, "elapsed": elapsed, "spmv_ms_total": spmv_ms, "spmv_count": spmv_n,
              "iterations": its, "stages": {k: float(np.mean(v)) for k, v in stages.items()},
              "t_input": t_input, "sell_padded": sell_padded, "sell_classes": sell_classes}
    return result


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(d, p, budget_nel):
    """The oracle's C + OpenMP restatement (oracle/tigar_oracle_c.c: the CSR algorithms PETSc AIJ runs on
    the CPU -- row-wise generateM, Gustavson PtAP + MatZeroRowsColumns, scatter-add M^T b, Jacobi-CG with
    PETSc's convergence test) on all host cores, on a bounded sample of the same workload (same d, p;
    fewer elements).  FE inputs A, b are generated beforehand (untimed, as on the GPU side)."""
    from oracle import tigar_oracle as O
    from oracle import tigar_oracle_c as OC
    nel = budget_nel
    # one thread per usable core, at most 32 (the sample is small; more threads only add barrier cost)
    OC.set_threads(min(32, OC.usable_cores()))
    t0 = time.perf_counter()
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    f = lambda x: np.sin(np.pi * x)
    # FE inputs from the device generator (identical matrices, seconds instead of half a minute of
    # scipy.kron); they are inputs of the baseline, not part of what is timed
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
    K = OC.extract_matrix(M, A, zd)
    rhs = OC.extract_vector(M, b, zd)
    t2 = time.perf_counter()
    U, its, _ = OC.cg_jacobi(K, rhs, rtol=1e-6)
    u = OC.spmv(M, U)
    t3 = time.perf_counter()
    total = t3 - t0
    ncp = s.getNcp()
    X, _ = O.fe_node_grid(s)
    err = float(np.max(np.abs(u - np.prod(np.sin(np.pi * X), axis=1))))
    return {"value": ncp / total, "unit": "DoF/s", "cores": OC.num_threads(), "kind": "port",
            "sample": "%dD p=%d %d^%d elements (%d DoFs): extract %.2fs, M^T A M + M^T b %.2fs, "
                      "CG(%d its)+prolongation %.2fs; C+OpenMP oracle on %d threads; max nodal error %.1e; "
                      "inputs %.2fs untimed"
                      % (d, p, nel, d, ncp, t1 - t0, t2 - t1, its, t3 - t2, OC.num_threads(), err, t_in)}


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
    ap.add_argument("--sub-planes", type=int, default=0, help="dof planes per streamed sub-slab (0 = auto)")
    ap.add_argument("--slab", type=int, default=-1, help="1: force the z-slab streaming path on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    cnt = counts(d, p, nel)

    # M + M^T + A + K resident at once needs ~ (2*nnzM + nnzA + 2*nnzK) * 12 B: stream in z-slabs
    # when that does not fit in ~60% of HBM (cfg3), or when asked to
    resident_bytes = 12.0 * (2 * cnt["nnzM"] + cnt["nnzA"] + 2 * cnt["nnzK"])
    use_slab = (args.gpus > 1 or world > 1 or args.slab == 1 or (args.slab != 0 and resident_bytes > 0.6 * 288e9))
    if use_slab:
        from bench_dist import run_distributed       # z-slab pipeline (+ RCCL when world > 1)
        res = run_distributed(args, d, p, nel, rank, world)
    else:
        res = run_single(args, d, p, nel)
    if rank != 0:
        return

    ms_per_step = 1e3 * res["elapsed"] / args.steps
    value = res["ncp"] / (res["elapsed"] / args.steps)
    spmv_avg_s = (res["spmv_ms_total"] / max(res["spmv_count"], 1)) * 1e-3
    alg_bytes = spmv_bytes(res["nnzK_local"] if "nnzK_local" in res else res["nnzK"],
                           res["ncp_local"] if "ncp_local" in res else res["ncp"])
    achieved = alg_bytes / spmv_avg_s / 1e9 if spmv_avg_s > 0 else 0.0
    # HBM traffic of the SpMV from the PMC passes (collected separately with rocprofv3 --pmc, as the
    # microarchitecture guide prescribes; committed under profiles/): only quoted for the workload
    # and GPU count it was measured on
    traffic = None
    sell = res.get("sell_padded", 0) > 0 and os.environ.get("TIGAR_SPMV_SELL", "1") != "0"
    kernel = "k_spmv_sell" if sell else "k_spmv_lane"
    pmc_file = os.path.join(ROOT, "profiles", "r1_spmv_pmc_summary.json")
    if wl == "cfg3" and max(args.gpus, world) == 1 and not args.nel and os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            traffic = pmc["hbm_bytes_per_launch"] if kernel in pmc["kernel"] else None
        except Exception:
            traffic = None
    # bytes the product kernel has to move in ITS format: the sliced copy streams 8 B per stored
    # position (values only; the offset dictionary is cache resident), class id + offset per slice,
    # x once and y once -- against SURVEY.md section 8(d)'s CSR figure (12 B per entry) in `achieved`
    ncp_l = res["ncp_local"] if "ncp_local" in res else res["ncp"]
    fmt_bytes = (8.0 * res["sell_padded"] + 12.0 * ((ncp_l + 63) // 64) + 16.0 * ncp_l) if sell else alg_bytes
    out = {
        "metric": "DoF/s (extraction + M^T A M + M^T b + CG solve + prolongation)",
        "value": value, "unit": "DoF/s", "n_gpus": max(args.gpus, world), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dD %d^%d elements p=%d B-spline Poisson, Q_p extraction, "
                               "Jacobi-CG rtol %.0e" % (wl, d, nel, d, p, args.rtol),
                   "dofs": res["ncp"], "fe_rows": cnt["rows_fe"], "nnz_M": cnt["nnzM"], "nnz_A": cnt["nnzA"],
                   "nnz_K": res["nnzK"], "cg_iterations": res["iterations"],
                   "stages_s": {k: round(v, 6) for k, v in res["stages"].items()},
                   "fe_input_generation_s": round(res["t_input"], 6),
                   "fe_input_inside_timed_region": bool(res.get("t_input_in_timed_region", False)),
                   "value_excluding_fe_input": res["ncp"] / max(1e-12, res["elapsed"] / args.steps
                                                                - (res["t_input"] if res.get("t_input_in_timed_region") else 0.0)),
                   "sub_planes": res.get("sub_planes"),
                   "parallelism": "z-slab x%d" % max(args.gpus, world)},
        "roofline": {"bound": "hbm", "kernel": "%s (K p in CG)" % kernel, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": "profiles/r1_spmv_pmc_summary.json (rocprofv3 --pmc "
                     "FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)" if traffic else None,
                     "launches": res["spmv_count"],
                     "avg_launch_ms": spmv_avg_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                     "format_bytes_per_launch": fmt_bytes,
                     "moved_GBps": fmt_bytes / spmv_avg_s / 1e9 if spmv_avg_s > 0 else 0.0,
                     "moved_frac_of_peak": fmt_bytes / spmv_avg_s / 1e9 / HBM_PEAK_GBS if spmv_avg_s > 0 else 0.0,
                     "note": ("achieved/frac are quoted on the CSR bytes of SURVEY.md 8(d) (12 B per entry) as the "
                              "contract asks; the kernel streams a sliced, pattern-compressed copy of the values "
                              "(8 B per stored position, no column indices), so it moves format_bytes_per_launch "
                              "= moved_GBps, moved_frac_of_peak of the 8 TB/s peak, and frac can exceed that")
                     if sell else None},
    }
    if not args.no_cpu_baseline and max(args.gpus, world) == 1:      # (rank 0 at N=1 only: the other ranks would wait for it)
        # a bounded sample: seconds of work on the 16 cores the GPU box grants (the A*M intermediate of
        # the Gustavson PtAP needs ~7 GB of host memory at p=3, 40^3 elements)
        cpu_nel = args.cpu_nel or ({2: 80, 3: 40, 4: 16}.get(p, 16) if d == 3 else min(nel, 256))
        out["cpu_baseline"] = cpu_baseline(d, p, cpu_nel)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
