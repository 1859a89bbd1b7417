"""Random 3-D patches with three displacement fields through the form-driven entry of linear elasticity (developer tool):
``assembleMatrix(ElasticityForm)`` -- nine field blocks, each a Kronecker sum fused into the first pass where the patch
qualifies, the general stages otherwise, plane-interleaved numbering when the operator is implicit -- against the oracle's
product with the assembled matrix (tIGAr/common.py:1206-1220 on EqualOrderSpline(3, ...), :1891-1914).

    python tests/fuzz/fuzz_elasticity.py [--seed S] [--cases N]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import tigar_oracle as O  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=40)
    a = ap.parse_args()
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    rng0 = np.random.default_rng(a.seed)
    bad = done = 0
    while done < a.cases:
        case = fz.draw_case(rng0, 4000)
        if case["d"] != 3 or len(set(case["ps"])) != 1:
            continue
        done += 1
        try:
            kvs = fz.knot_vectors(case, B.uniformKnots)
            gen = t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh(case["ps"], kvs))
            sp0 = gen.getScalarSpline(0)
            for f in range(3):
                for k in range(3):
                    if case["kinds"][k] != "periodic":
                        gen.addZeroDofs(f, sp0.getSideDofs(k, 0))
            spline = t.ExtractedSpline(gen, 2 * max(case["ps"]))
            form = F.ElasticityForm(1.3, 0.7)
            so = O.BSpline(case["ps"], kvs)
            Mo = O.generate_M_tensor(so, nfields=3)
            V3 = t.ExtractedSpline(t.EqualOrderSpline(3, B.ExplicitBSplineControlMesh(case["ps"], kvs)), 2).V
            A = form.assemble_matrix(V3).to_scipy().tocsr()
            uks = [np.asarray(s1.uniqueKnots, dtype=float) for s1 in so.splines]
            Ao = O.elasticity_fe_system(uks, case["ps"][0], 1.3, 0.7)
            assert abs(A - Ao).max() <= 1e-12 * abs(Ao).max(), "FE matrix: %g" % (abs(A - Ao).max() / abs(Ao).max())
            zd = list(spline.zeroDofs)
            Ko = O.extract_matrix(Mo, A, zd, diag=1.0)
            idx = np.asarray(spline.localDofIndices(), dtype=np.int64)
            if not np.array_equal(idx, np.arange(Ko.shape[0])):
                Ko = Ko.tocsr()[idx][:, idx].tocsr()
            Ko.sort_indices()
            K = spline.assembleMatrix(form).to_scipy()
            K.sort_indices()
            e = abs(K - Ko).max() / abs(Ko).max()
            assert e <= 1e-12, "values %g" % e
            assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices), "pattern (%d / %d)" % (K.nnz, Ko.nnz)
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print(json.dumps({"failed": done - 1, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200]), "case": case}), flush=True)
    print(json.dumps({"cases": a.cases, "failed": bad, "seed": a.seed}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
