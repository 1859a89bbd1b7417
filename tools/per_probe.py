import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, common as tc
from oracle import tigar_oracle as O
def run(periodic, box):
    os.environ["TIGAR_PTAP_BOX"] = box
    d, p, nel = 3, 2, int(os.environ.get("NEL", "9"))
    kv = [B.uniformKnots(p, 0., 1., nel, periodic and k == 0) for k in range(d)]
    gen = t.EqualOrderSpline(tc.selfcomm, 1, B.ExplicitBSplineControlMesh([p] * d, kv))
    sp0 = gen.getScalarSpline(0)
    for direction in range(1 if periodic else 0, d):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p, comm=tc.selfcomm)
    try:
        K = spline.assembleMatrix(F.LaplaceForm(), diag=1.5).to_scipy()
        A = F.LaplaceForm().assemble_matrix(spline.V).to_scipy()
        M = gen.M.to_scipy()
        Ko = O.extract_matrix(M, A, list(spline.zeroDofs), diag=1.5)
        print("periodic", periodic, "box", box, "ok, max diff", abs(K - Ko).max(), "nnz", K.nnz, Ko.nnz)
    except Exception as e:
        print("periodic", periodic, "box", box, "FAILED:", str(e)[:200])
for per in (False, True):
    for box in ("1", "0"):
        run(per, box)
