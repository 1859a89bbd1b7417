"""M^T A M for an FE matrix with a few couplings outside the element-coupling pattern: pattern split against the general
line kernels on the whole matrix (developer tool).  usage: split_bench.py p nel n_extra"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, device as dev, forms as F
p, nel, nx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kvs = [B.uniformKnots(p, 0., 1., nel) for _ in range(3)]
gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kvs))
spline = t.ExtractedSpline(gen, 2 * p)
A = F.LaplaceForm().assemble_matrix(spline.V)
n = A.shape[0]
rng = np.random.default_rng(0)
R = sp.csr_matrix((rng.standard_normal(nx), (rng.integers(0, n, nx), rng.integers(0, n, nx))), shape=(n, n))
R.sum_duplicates(); R.sort_indices()
A2 = A.add(dev.DeviceCSR.from_scipy(R))
print("A: %.2f GB, %d entries added outside the pattern" % (12 * A.nnz / 1e9, A2.nnz - A.nnz), flush=True)
for name, env in (("on the pattern (reference)", None), ("split", None), ("general line kernels", "0")):
    M = A if name.startswith("on") else A2
    if env is not None:
        os.environ["TIGAR_PTAP_SPLIT"] = env
    ts = []
    for _ in range(3):
        dev.sync(); t0 = time.perf_counter()
        K = spline.extractMatrix(M)
        dev.sync(); ts.append(time.perf_counter() - t0)
        nnz = K.nnz
        del K
    os.environ.pop("TIGAR_PTAP_SPLIT", None)
    print("%-28s %.1f ms  (nnz(K) = %d)" % (name, 1e3 * min(ts), nnz), flush=True)
# stage times of the split path
from tigar_amd.tensorptap import TensorPtAP
plan = TensorPtAP.for_extraction(spline._kron)
def T(f):
    dev.sync(); t0 = time.perf_counter(); r = f(); dev.sync(); return r, 1e3 * (time.perf_counter() - t0)
(on, off), t1 = T(lambda: plan.split(A2))
nz = spline._kron.nfe[-1]; kz = spline._kron.ncp[-1]
Kt, t2 = T(lambda: plan.zstage([plan.planes(on, 0, 0, nz)], 0, kz))
Kr, t3 = T(lambda: dev.ptap_numeric(dev.ptap_symbolic(off, spline.M, spline.MT), off, spline.M, spline.MT))
K, t4 = T(lambda: Kt.add(Kr))
print("split %.1f ms, passes on the pattern part %.1f ms, hash product of the remainder %.1f ms (nnz %d), add %.1f ms" % (t1, t2, t3, Kr.nnz, t4))
