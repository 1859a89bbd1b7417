"""PtAP with an extraction operator that is NOT a Kronecker product (developer tool / profile source).

A synthetic unstructured spline space in the layout of the reference's Rhino T-splines (tIGAr/RhinoTSplines.py:37-137):
the FE mesh consists of DISCONNECTED bicubic cells (16 Lagrange nodes each, numbered cell after cell), every cell lists
its 16 spline functions and their values at its nodes (uniform bicubic B-spline functions, exact zeros at the knots dropped
by generateM's filter), and the functions carry a RANDOM global numbering -- no lattice, no runs of consecutive columns,
nothing the Kronecker / box / line / tensor kernels could use.  A is what dolfin assembles on such a mesh: one dense 16 x 16
block per cell (here: SPD blocks with a deterministic perturbation per cell).

    python tools/general_ptap_bench.py [cells_per_side=256] [reps=5] [extra=0.0]

extra > 0: that fraction of the FE rows additionally gets one coupling to a node of ANOTHER cell (contact / penalty terms
added by hand, demos/kl-shell-svk/reef-knot.py:455-467); the product then runs as extractMatrix runs it -- split on the
device into cell blocks + remainder, cell-block product + general kernels on the remainder, sum on the union pattern.

Prints one JSON line: sizes, PtAP time (first call = symbolic + numeric, later calls = numeric on the plan), SURVEY 8(d)'s
algorithmic bytes / time against the 8 TB/s peak, and the check K x = M^T (A (M x)) on a random x.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev  # noqa: E402


def cubic_bspline_at(t):
    """values of the four uniform cubic B-spline functions that live on a knot span, at local coordinate t"""
    return np.array([(1 - t) ** 3, 3 * t ** 3 - 6 * t ** 2 + 4, -3 * t ** 3 + 3 * t ** 2 + 3 * t + 1, t ** 3]) / 6.0


def build(n, seed=0):
    rng = np.random.default_rng(seed)
    L1 = np.array([cubic_bspline_at(t) for t in (0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0)])         # [node][function]
    L = np.kron(L1, L1)                                                                     # 16 x 16, y outer
    nf = n + 3
    perm = rng.permutation(nf * nf).astype(np.int64)                                        # scrambled dof numbering
    ex, ey = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    ex, ey = ex.ravel(), ey.ravel()
    a = np.arange(4)
    fx = (ex[:, None, None] + a[None, None, :])                                             # [cell][ay][ax]
    fy = (ey[:, None, None] + a[None, :, None])
    dof = perm[(fx + nf * fy).reshape(-1, 16)]                                              # [cell][16]
    ncell = n * n
    rows = (np.arange(ncell * 16).reshape(ncell, 16, 1) + np.zeros((1, 1, 16), dtype=np.int64))
    cols = dof[:, None, :] + np.zeros((1, 16, 1), dtype=np.int64)
    vals = np.broadcast_to(L[None, :, :], (ncell, 16, 16))
    keep = np.abs(vals) > 1e-15                                                             # generateM's filter
    M = sp.csr_matrix((vals[keep], (rows[keep], cols[keep])), shape=(ncell * 16, nf * nf))
    M.sort_indices()
    # cell matrices: a fixed SPD block, scaled and perturbed per cell
    B = rng.standard_normal((16, 16))
    B = B @ B.T + 16 * np.eye(16)
    scale = 1.0 + 0.5 * np.sin(0.37 * np.arange(ncell))
    blocks = scale[:, None, None] * B[None, :, :]
    r = (np.arange(ncell * 16).reshape(ncell, 16, 1) + np.zeros((1, 1, 16), dtype=np.int64)).ravel()
    c = (np.arange(ncell * 16).reshape(ncell, 1, 16) + np.zeros((1, 16, 1), dtype=np.int64)).ravel()
    A = sp.csr_matrix((blocks.ravel(), (r, c)), shape=(ncell * 16, ncell * 16))
    A.sort_indices()
    return M, A


def with_extras(n, reps, extra, M, A):
    from tigar_amd.cellptap import CellBlockPtAP, split_cells, cell_size_with_extras
    rng = np.random.default_rng(7)
    nfe = A.shape[0]
    ne = max(1, int(round(extra * nfe)))
    r = rng.choice(nfe, size=ne, replace=False)
    c = (r + 16 * rng.integers(1, nfe // 16, size=ne)) % nfe              # a node of another cell
    E = sp.csr_matrix((rng.standard_normal(ne), (r, c)), shape=A.shape)
    Ax = (A + E).tocsr()
    Ax.sort_indices()
    Md, Ad = dev.DeviceCSR.from_scipy(M), dev.DeviceCSR.from_scipy(Ax)
    MT = Md.transpose()
    dev.sync()
    # the general kernels on the whole matrix (what round 4 did with such a matrix)
    plan = dev.ptap_symbolic(Ad, Md, MT)
    Kg = dev.ptap_numeric(plan, Ad, Md, MT)
    dev.sync()
    tg = []
    for _ in range(reps):
        t0 = time.perf_counter()
        Kg = dev.ptap_numeric(plan, Ad, Md, MT)
        dev.sync()
        tg.append(time.perf_counter() - t0)
    # the split path, as ExtractedSpline.extractMatrix runs it: cell blocks read in place + remainder restricted to its rows
    from tigar_amd.cellptap import remainder_product
    assert cell_size_with_extras(Ad) == 16
    cplan = CellBlockPtAP(Md, 16)
    stages = {"cells_and_split": [], "remainder": [], "add": []}
    rcache = {}

    def once():
        t0 = time.perf_counter()
        KD, R = cplan.ptap_extras(Ad)
        dev.sync()
        t1 = time.perf_counter()
        KR = remainder_product(R, Md, rcache)
        dev.sync()
        t2 = time.perf_counter()
        K = KD.add(KR)
        dev.sync()
        t3 = time.perf_counter()
        return K, (t1 - t0, t2 - t1, t3 - t2)
    t0 = time.perf_counter()
    K, _ = once()
    first_ms = time.perf_counter() - t0
    ts = []
    for _ in range(reps):
        K, tt = once()
        ts.append(sum(tt))
        for k_, v in zip(stages, tt):
            stages[k_].append(v)
    Ks, Kgs = K.to_scipy().tocsr(), Kg.to_scipy().tocsr()
    Ks.sort_indices(), Kgs.sort_indices()
    x = np.random.default_rng(1).standard_normal(M.shape[1])
    yref = M.T @ (Ax @ (M @ x))
    err = float(np.max(np.abs(Ks @ x - yref)) / np.max(np.abs(yref)))
    algo = 12.0 * Ax.nnz + 24.0 * M.nnz + 12.0 * K.nnz + 8.0 * (2 * M.shape[0] + 2 * M.shape[1])
    best = min(ts)
    print(json.dumps({
        "workload": "non-Kronecker M: %d x %d disconnected bicubic cells, scrambled dof numbering, + %d couplings between cells "
                    "(%.2f %% of the FE rows)" % (n, n, ne, 100.0 * ne / nfe),
        "fe_rows": int(M.shape[0]), "dofs": int(M.shape[1]), "nnz_M": int(M.nnz), "nnz_A": int(Ax.nnz), "nnz_K": int(K.nnz),
        "algorithmic_bytes": algo, "bytes_definition": "SURVEY.md 8d: 12 nnz(A) + 24 nnz(M) + 12 nnz(K) + row pointers",
        "general_kernels_whole_matrix_ms": 1e3 * min(tg), "general_kernels_frac_of_hbm_peak": algo / min(tg) / 8e12,
        "split_path_ms": 1e3 * best, "split_path_achieved_GBps": algo / best / 1e9, "split_path_frac_of_hbm_peak": algo / best / 8e12,
        "split_path_stages_ms": {k_: 1e3 * min(v) for k_, v in stages.items()},
        "split_path_first_call_ms": 1e3 * first_ms,
        "pattern_equals_general_kernels": bool(np.array_equal(Ks.indices, Kgs.indices)),
        "max_rel_diff_vs_general_kernels": float(abs(Ks - Kgs).max() / abs(Kgs).max()), "rel_error_Kx_vs_MtAMx": err}))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    extra = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    M, A = build(n)
    if extra > 0.0:
        return with_extras(n, reps, extra, M, A)
    Md, Ad = dev.DeviceCSR.from_scipy(M), dev.DeviceCSR.from_scipy(A)
    MT = Md.transpose()
    dev.sync()

    def first():
        return dev.ptap_numeric(dev.ptap_symbolic(Ad, Md, MT), Ad, Md, MT)
    K = first()
    dev.sync()
    t_first = []
    for _ in range(reps):
        t0 = time.perf_counter()
        K = first()
        dev.sync()
        t_first.append(time.perf_counter() - t0)
    plan = dev.ptap_symbolic(Ad, Md, MT)
    K = dev.ptap_numeric(plan, Ad, Md, MT)
    dev.sync()
    t_again = []
    for _ in range(reps):
        t0 = time.perf_counter()
        K = dev.ptap_numeric(plan, Ad, Md, MT)
        dev.sync()
        t_again.append(time.perf_counter() - t0)
    # the cell-block product (tigar_amd/cellptap.py): plan = symbolic half, built once per extraction operator on the host
    cells = {}
    if os.environ.get("TIGAR_PTAP_CELLS", "1") != "0":
        from tigar_amd.cellptap import CellBlockPtAP
        t0 = time.perf_counter()
        cplan = CellBlockPtAP(Md, 16)
        dev.sync()
        cells["plan_s"] = time.perf_counter() - t0
        Kc = cplan.ptap(Ad)
        dev.sync()
        tc = []
        for _ in range(reps):
            t0 = time.perf_counter()
            Kc = cplan.ptap(Ad)
            dev.sync()
            tc.append(time.perf_counter() - t0)
        cells["ptap_ms"] = 1e3 * min(tc)
        Kcs, Kgs = Kc.to_scipy().tocsr(), K.to_scipy().tocsr()
        Kcs.sort_indices(), Kgs.sort_indices()
        cells["pattern_equals_general_kernels"] = bool(np.array_equal(Kcs.indices, Kgs.indices))
        cells["max_rel_diff_vs_general_kernels"] = float(abs(Kcs - Kgs).max() / abs(Kgs).max())
    x = np.random.default_rng(1).standard_normal(M.shape[1])
    y = K.to_scipy() @ x
    yref = M.T @ (A @ (M @ x))
    err = float(np.max(np.abs(y - yref)) / np.max(np.abs(yref)))
    K2 = dev.ptap_numeric(plan, Ad, Md, MT).to_scipy()
    same = bool(np.array_equal(K2.data.view(np.int64), K.to_scipy().data.view(np.int64)))
    nnzK = K.nnz
    algo = 12.0 * A.nnz + 24.0 * M.nnz + 12.0 * nnzK + 8.0 * (2 * M.shape[0] + 2 * M.shape[1])
    out = {"workload": "non-Kronecker M: %d x %d disconnected bicubic cells, scrambled dof numbering" % (n, n),
           "fe_rows": int(M.shape[0]), "dofs": int(M.shape[1]), "nnz_M": int(M.nnz), "nnz_A": int(A.nnz), "nnz_K": int(nnzK),
           "kernels": "wave" if os.environ.get("TIGAR_PTAP_WAVE", "1") != "0" else "workgroup",
           "ptap_first_call_ms": 1e3 * min(t_first), "ptap_numeric_on_plan_ms": 1e3 * min(t_again),
           "algorithmic_bytes": algo, "achieved_GBps_first_call": algo / min(t_first) / 1e9,
           "achieved_GBps_on_plan": algo / min(t_again) / 1e9,
           "frac_of_hbm_peak_on_plan": algo / min(t_again) / 8e12,
           "bytes_definition": "SURVEY.md 8d: 12 nnz(A) + 24 nnz(M) + 12 nnz(K) + row pointers",
           "rel_error_Kx_vs_MtAMx": err, "bit_reproducible": same}
    if cells:
        out["cell_block_product"] = dict(cells, achieved_GBps=algo / (cells["ptap_ms"] * 1e-3) / 1e9,
                                         frac_of_hbm_peak=algo / (cells["ptap_ms"] * 1e-3) / 8e12,
                                         note="K = sum_c S_c^T (M_c^T A_c M_c) S_c: dense element matrices, look-up only to merge them; "
                                              "plan_s = the symbolic half on the host, once per extraction operator")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
