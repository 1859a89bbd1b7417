"""Times the products of a CG solve on K = M^T A M of a 3-D Poisson problem: sliced copy (tg_sell.hip) vs half-storage copy
(tg_symgrid.hip).  usage: python tools/symgrid_bench.py [nel] [p] [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nel = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = sys.argv[3] if len(sys.argv) > 3 else None
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev
    kv = [B.uniformKnots(p, 0., 1., nel)] * 3
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
    s0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    K = spline.assembleMatrix(F.LaplaceForm())
    rhs = spline.assembleVector(F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * 3, scale=3 * np.pi ** 2))
    n = K.shape[0]
    _, info = K.mult_symgrid()
    res = {"nel": nel, "p": p, "rows": n, "nnz": K.nnz, "symgrid": info}
    for mode in ("0", "1"):
        os.environ["TIGAR_SPMV_SYM"] = mode
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-6
        best = None
        for rep in range(3):
            U = dev.DeviceVector(n)
            dev.prof_reset()
            dev.sync()
            t0 = time.perf_counter()
            its = ks.solve(K, U, rhs)
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
