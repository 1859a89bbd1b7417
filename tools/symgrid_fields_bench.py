"""Times the products of a CG solve on the K of linear elasticity (three displacement fields on one 3-D tensor basis): sliced
copy (tg_sell.hip) vs the half-storage copy of the nine-block matrix (tg_symgrid.hip, several fields).
usage: python tools/symgrid_fields_bench.py [nel] [p] [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nel = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = sys.argv[3] if len(sys.argv) > 3 else None
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev
    kv = [B.uniformKnots(p, 0., 1., nel)] * 3
    gen = t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh([p] * 3, kv))
    s0 = gen.getScalarSpline(0)
    for f in range(3):
        gen.addZeroDofs(f, s0.getSideDofs(0, 0))
    spline = t.ExtractedSpline(gen, 2 * p)
    t0 = time.perf_counter()
    K = spline.assembleMatrix(F.ElasticityForm(1.3, 0.7))
    dev.sync()
    n = K.shape[0]
    print("K: %d rows, %d entries (%.1f GB as CSR), assembled + extracted in %.2f s" % (n, K.nnz, 12e-9 * K.nnz, time.perf_counter() - t0), flush=True)
    b = np.zeros(n)
    b[2 * (n // 3):] = -1.0
    rhs = dev.DeviceVector(data=b)
    rhs.zero_entries(np.asarray(sorted(spline.zeroDofs), dtype=np.int64), 0)
    t0 = time.perf_counter()
    _, info = K.mult_symgrid()
    dev.sync()
    res = {"nel": nel, "p": p, "rows": n, "nnz": K.nnz, "symgrid": info, "plan_and_check_s": time.perf_counter() - t0}
    for mode in ("0", "1"):
        os.environ["TIGAR_SPMV_SYM"] = mode
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-6
        ks.parameters["maximum_iterations"] = 300
        best = None
        for rep in range(2):
            U = dev.DeviceVector(n)
            dev.prof_reset()
            dev.sync()
            t0 = time.perf_counter()
            try:
                its = ks.solve(K, U, rhs)
            except RuntimeError:           # (the iteration limit: the timing of the products is what this tool is for)
                its = -1
            dev.sync()
            dt = time.perf_counter() - t0
            ms, cnt = dev.prof_get(0)
            row = {"solve_s": dt, "iterations": its, "product_ms": ms / max(cnt, 1), "products": cnt}
            if best is None or dt < best["solve_s"]:
                best = row
        res["sym" if mode == "1" else "sell"] = best
        print(mode, best, flush=True)
    if info:
        res["sym"]["GBps_values"] = info["value_bytes"] / res["sym"]["product_ms"] / 1e6
    res["sell"]["GBps_values"] = 8.0 * K.nnz / res["sell"]["product_ms"] / 1e6
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


main()
