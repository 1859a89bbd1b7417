"""extractMatrix on a 2-D div-conforming B-spline space (BSplineCompat("RT"), the space of demos/taylor-green/taylor-green-2d.py):
the four field blocks K_fg = M_f^T A_fg M_g through the 2-D line walks with different row / column bases (round 6,
``TensorPtAP2D.for_pair``) against the general kernels (TIGAR_PTAP_TENSOR=0), on an assembled block matrix on the common Q_P grid.

    python tools/rt2d_bench.py [k=1] [nel=512] [out.json]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t  # noqa: E402
from tigar_amd import BSplines as B, common as tc, device as dev, forms as F  # noqa: E402
from tigar_amd.compatibleSplines import BSplineCompat  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nel = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    out = sys.argv[3] if len(sys.argv) > 3 else None
    degs = [k, k]
    kv = [B.uniformKnots(k, 0., 1., nel) for _ in range(2)]
    gen = BSplineCompat(tc.selfcomm, B.ExplicitBSplineControlMesh(degs, kv), "RT", degs)
    for field in range(2):
        sp_f = gen.getFieldSpline(field)
        for side in (0, 1):
            gen.addZeroDofs(field, sp_f.getSideDofs(field, side))
    spline = t.ExtractedSpline(gen, 2 * (k + 1))
    g = spline.V.grids[0]
    V1 = type(spline.V)([g], spline.V.element)
    L = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
    rng = np.random.default_rng(0)
    blk = [[None, None], [None, None]]
    for a in range(2):
        for b in range(2):
            Bm = L.copy()
            Bm.data = Bm.data + 0.2 * rng.standard_normal(Bm.nnz)
            blk[a][b] = Bm
    A = dev.DeviceCSR.from_scipy(sps.bmat(blk, format="csr"))

    def timed(n=5):
        K = spline.extractMatrix(A)
        dev.sync()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            K = spline.extractMatrix(A)
            dev.sync()
            ts.append(time.perf_counter() - t0)
        return K, min(ts) * 1e3
    dev.prof_reset()
    K, ms_walks = timed()
    walks = dev.prof_get(5)[1]
    os.environ["TIGAR_PTAP_TENSOR"] = "0"
    Kg, ms_general = timed(2)
    os.environ.pop("TIGAR_PTAP_TENSOR")
    x = dev.DeviceVector(data=rng.standard_normal(K.shape[0]))
    y1, y2 = K.mult(x).get_local(), Kg.mult(x).get_local()
    M = gen.M
    bytes_8d = 12.0 * A.nnz + 24.0 * M.nnz + 12.0 * K.nnz
    res = {"space": "BSplineCompat RT, k = %d, %d x %d elements (Q_%d node grid)" % (k, nel, nel, k + 1), "fe_rows": A.shape[0],
           "dofs": K.shape[0], "nnz_A": A.nnz, "nnz_M": M.nnz, "nnz_K": K.nnz, "final_passes_of_the_walks": int(walks),
           "extractMatrix_pair_walks_ms": ms_walks, "extractMatrix_general_kernels_ms": ms_general,
           "bytes_8d": bytes_8d, "pair_walks_frac_of_8TBps": bytes_8d / (ms_walks * 1e-3) / 8e12,
           "general_frac_of_8TBps": bytes_8d / (ms_general * 1e-3) / 8e12,
           "same_entries": bool(K.nnz == Kg.nnz), "rel_diff_Kx": float(np.max(np.abs(y1 - y2)) / np.max(np.abs(y2)))}
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
