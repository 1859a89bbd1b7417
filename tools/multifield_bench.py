"""M^T A M for three fields on one 3-D tensor basis: the block-by-block tensor path against the general kernels
(developer tool).  usage: multifield_bench.py p nel"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, device as dev, forms as F
p, nel, nF = int(sys.argv[1]), int(sys.argv[2]), 3
kvs = [B.uniformKnots(p, 0., 1., nel) for _ in range(3)]
gen = t.EqualOrderSpline(nF, B.ExplicitBSplineControlMesh([p] * 3, kvs))
sp0 = gen.getScalarSpline(0)
for f in range(nF):
    gen.addZeroDofs(f, sp0.getSideDofs(0, 0))
spline = t.ExtractedSpline(gen, 2 * p)
gen1 = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kvs))
A1 = F.LaplaceForm().assemble_matrix(gen1.V)
A = dev.csr_from_blocks([[A1] * nF for _ in range(nF)])
print("A: %d rows, %.2f GB; dofs %d" % (A.shape[0], 12 * A.nnz / 1e9, spline.M.shape[1]), flush=True)
for name, env in (("field blocks (tensor passes)", None), ("general kernels", "0")):
    if env is not None:
        os.environ["TIGAR_PTAP_FACTORED"] = env
    ts = []
    for _ in range(3):
        dev.sync(); t0 = time.perf_counter()
        K = spline.extractMatrix(A)
        dev.sync(); ts.append(time.perf_counter() - t0)
        nnz = K.nnz
        del K
    os.environ.pop("TIGAR_PTAP_FACTORED", None)
    print("%-30s %.1f ms  (nnz(K) = %d)" % (name, 1e3 * min(ts), nnz), flush=True)
