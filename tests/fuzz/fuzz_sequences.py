"""Sequences of DIFFERENT FE matrices through one ExtractedSpline (developer tool): the plans the product caches between calls
-- symbolic products keyed by shape and nnz, tensor-pattern plans, fold plans keyed by a checksum of K_u's pattern, cell
plans -- must never serve a matrix they were not made for (the reference recomputes the symbolic product on every call,
tIGAr/common.py:1194-1195).  Per random patch: Laplace, random values on the pattern, two matrices with the SAME number of
hand-added couplings at different places, Laplace again, mass; each product against the oracle.

    python tests/fuzz/fuzz_sequences.py [--seed S] [--cases N]"""
import argparse
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import tigar_oracle as O  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=60)
    a = ap.parse_args()
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    rng0 = np.random.default_rng(a.seed)
    bad = 0
    for i in range(a.cases):
        case = fz.draw_case(rng0, 20000)
        case["nfields"] = 1 if case["nfields"] == 1 else 2
        nf = case["nfields"]
        try:
            kvs = fz.knot_vectors(case, B.uniformKnots)
            gen = t.EqualOrderSpline(nf, B.ExplicitBSplineControlMesh(case["ps"], kvs))
            so = O.BSpline(case["ps"], kvs)
            gen.addZeroDofs(0, [0, 1])
            spline = t.ExtractedSpline(gen, 2 * max(case["ps"]))
            V1 = spline.V if nf == 1 else t.ExtractedSpline(t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh(case["ps"], kvs)), 2).V
            L = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
            Mm = F.MassForm().assemble_matrix(V1).to_scipy().tocsr()
            if nf > 1:
                L, Mm = sp.block_diag([L] * nf, format="csr"), sp.block_diag([Mm] * nf, format="csr")
            Mo = O.generate_M_tensor(so, nfields=nf)
            rng = np.random.default_rng(case["val_seed"])
            R = L.copy()
            R.data = rng.standard_normal(R.nnz)
            n = L.shape[0]

            def extra(seed):
                r2 = np.random.default_rng(seed)
                for _ in range(500):
                    E = sp.csr_matrix((r2.standard_normal(3), (r2.integers(0, n, 3), r2.integers(0, n, 3))), shape=L.shape)
                    X = (R + E).tocsr()
                    if X.nnz == R.nnz + 3:          # three NEW positions: every variant has the same nnz
                        return X
                return None                         # (a patch of one element: the FE matrix is dense, nothing to add)
            seq = [("laplace", L), ("random", R), ("extra a", extra(1)), ("extra b", extra(2)), ("extra c", extra(3)),
                   ("laplace again", L), ("mass", Mm), ("extra a again", extra(1))]
            seq = [(nm, X) for nm, X in seq if X is not None]
            zd = list(spline.zeroDofs)
            idx = np.asarray(spline.localDofIndices(), dtype=np.int64)       # (several implicit fields: plane by plane)
            for name, A in seq:
                A = A.tocsr()
                A.sort_indices()
                K = spline.extractMatrix(A, diag=2.0).to_scipy()
                Ko = O.extract_matrix(Mo, A, zd, diag=2.0)
                if not np.array_equal(idx, np.arange(Ko.shape[0])):
                    Ko = Ko.tocsr()[idx][:, idx].tocsr()
                K.sort_indices(), Ko.sort_indices()
                e = abs(K - Ko).max() / abs(Ko).max()
                assert e <= 1e-12, "%s: values %g" % (name, e)
                assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices), "%s: pattern" % name
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print(json.dumps({"failed": i, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200]), "case": case}), flush=True)
    print(json.dumps({"cases": a.cases, "failed": bad, "seed": a.seed}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
