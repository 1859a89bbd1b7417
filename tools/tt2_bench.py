"""developer tool: wall time per call of the 2-D tensor-pattern PtAP (tg_tensor2_ptap) at cfg4 / cfg5 sizes, through the
plan object and through ExtractedSpline.extractMatrix; TIGAR_TT2_ECH_X / _Y select the piece lengths of the walks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, device as dev
from tigar_amd.tensorptap import TensorPtAP2D

def run(p, nel, nF, reps=40):
    kv = [B.uniformKnots(p, -1., 1., nel)] * 2
    gen = t.EqualOrderSpline(nF, B.ExplicitBSplineControlMesh([p, p], kv))
    for f in range(nF):
        s0 = gen.getScalarSpline(f)
        for d in range(2):
            for side in (0, 1):
                gen.addZeroDofs(f, s0.getSideDofs(d, side, nLayers=2))
    spline = t.ExtractedSpline(gen, 2 * p)
    if nF == 1:
        A = F.BiharmonicForm().assemble_matrix(spline.V)
        kx = spline._kron
    else:
        import scipy.sparse as sp
        pat = F.LaplaceForm().assemble_matrix(t.TensorFunctionSpace([gen.getScalarSpline(0).generateMesh(degree=p)], "Lagrange")).to_scipy()
        A = dev.DeviceCSR.from_scipy(sp.bmat([[pat] * nF for _ in range(nF)], format="csr"))
        kx = spline._kron_scalar
    plan = TensorPtAP2D.for_extraction(kx, nF)
    zd = spline.zeroDofs
    for name, fn in (("plan.ptap", lambda: plan.ptap(A, zd, 1.0)), ("extractMatrix", lambda: spline.extractMatrix(A))):
        for _ in range(3):
            K = fn()
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            K = fn()
        dev.sync()
        print("p=%d nel=%d nF=%d %-14s %.3f ms per call (nnz K %d) ech %s/%s" % (p, nel, nF, name, 1e3 * (time.perf_counter() - t0) / reps, K.nnz,
              os.environ.get("TIGAR_TT2_ECH_X", "auto"), os.environ.get("TIGAR_TT2_ECH_Y", "auto")), flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("cfg4", "both"):
        run(4, 256, 1)
    if which in ("cfg5", "both"):
        run(3, 128, 3)
