"""extractMatrix on a patch with periodic directions (developer tool / profile source): the tensor line walks on the unwrapped
space + the fold K = R^T K_u R (tigar_amd/kronptap.py: KronExtraction.unwrapped / fold) against the general stages that such
patches took before round 4 (TIGAR_PTAP_UNWRAP=0).

    python tools/periodic_ptap_bench.py [p=3] [nel=64] [periodic directions, e.g. 01] [reps=3]

Prints one JSON line: sizes, time per extractMatrix call on either path, SURVEY 8(d)'s algorithmic bytes over the time, and
the largest difference between the two results."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t  # noqa: E402
from tigar_amd import BSplines as B, forms as F, device as dev  # noqa: E402


def main():
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nel = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    per = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "01")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    d = 3
    kv = [B.uniformKnots(p, 0., 1., nel, k in per) for k in range(d)]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, kv))
    sp0 = gen.getScalarSpline(0)
    for k in range(d):
        if k not in per:
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(k, side))
    out = {"workload": "3-D %d^3 elements p=%d, periodic directions %s" % (nel, p, per)}
    res = {}
    for mode in ("1", "0"):
        os.environ["TIGAR_PTAP_UNWRAP"] = mode
        spline = t.ExtractedSpline(gen, 2 * p)
        A = F.LaplaceForm().assemble_matrix(spline.V)
        dev.prof_reset()
        K = spline.extractMatrix(A, diag=1.0)
        dev.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            K = spline.extractMatrix(A, diag=1.0)
            dev.sync()
            ts.append(time.perf_counter() - t0)
        name = "walks_on_unwrapped_space_then_fold" if mode == "1" else "general_stages"
        algo = 12.0 * A.nnz + 24.0 * spline.M.nnz + 12.0 * K.nnz + 8.0 * (2 * A.shape[0] + 2 * K.shape[0])
        res[name] = {"ms": 1e3 * min(ts), "tensor_walk_stages": int(dev.prof_get(5)[1]), "achieved_GBps": algo / min(ts) / 1e9,
                     "frac_of_hbm_peak": algo / min(ts) / 8e12}
        out.update({"fe_rows": int(A.shape[0]), "dofs": int(K.shape[0]), "nnz_A": int(A.nnz), "nnz_M": int(spline.M.nnz),
                    "nnz_K": int(K.nnz), "algorithmic_bytes": algo})
        res[name + "_K"] = K
        del A, spline
    Ka, Kb = res.pop("walks_on_unwrapped_space_then_fold_K"), res.pop("general_stages_K")
    x = dev.DeviceVector(data=np.random.default_rng(0).standard_normal(Ka.shape[0]))
    ya, yb = dev.DeviceVector(Ka.shape[0]), dev.DeviceVector(Ka.shape[0])
    Ka.mult(x, ya), Kb.mult(x, yb)
    out["same_pattern_size"] = bool(Ka.nnz == Kb.nnz)
    out["max_rel_diff_Kx"] = float(np.max(np.abs(ya.get_local() - yb.get_local())) / np.max(np.abs(yb.get_local())))
    out.update(res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
