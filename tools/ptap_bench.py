"""micro-benchmark of the PtAP / transpose / extraction kernels (developer tool)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm

d, p, nel = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 96
reps = 3
cm = ExplicitBSplineControlMesh([p]*d, [uniformKnots(p, 0., 1., nel)]*d)
basis = cm.getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
A = LaplaceForm().assemble_matrix(V)
def T(f, n=reps):
    dev.sync(); ts=[]
    for _ in range(n):
        t=time.perf_counter(); r=f(); dev.sync(); ts.append(time.perf_counter()-t)
    return min(ts)*1e3, r
t, M = T(lambda: dev.extract_csr_tensor(basis.splines, grid.axes, 0, basis.getNcp(), 1e-15))
print("extract   %8.2f ms  nnz %d  -> %.0f GB/s" % (t, M.nnz, (12*M.nnz+8*M.shape[0])/t/1e6))
def tr():
    M._T=None; return M.transpose()
t, MT = T(tr)
print("transpose %8.2f ms" % t)
t, plan = T(lambda: dev.ptap_symbolic(A, M, MT))
print("ptap probe%8.2f ms" % t)
def num_first():
    pl = dev.ptap_symbolic(A, M, MT); return dev.ptap_numeric(pl, A, M, MT)
t, K = T(num_first)
print("ptap first%8.2f ms (probe+bump+reorder)  nnzK %d" % (t, K.nnz))
pl = dev.ptap_symbolic(A, M, MT); K = dev.ptap_numeric(pl, A, M, MT)
t, K2 = T(lambda: dev.ptap_numeric(pl, A, M, MT))
print("ptap again%8.2f ms (placed)" % t)
x = dev.DeviceVector(data=np.random.default_rng(0).standard_normal(K.shape[0])); y = dev.DeviceVector(K.shape[0])
K.mult(x, y)
t, _ = T(lambda: [K.mult(x, y) for _ in range(20)])
b = 12*K.nnz + 4*(K.shape[0]+1) + 16*K.shape[0]
print("spmv K    %8.3f ms -> %.0f GB/s algorithmic" % (t/20, b/(t/20)/1e6))
xm = dev.DeviceVector(data=np.ones(M.shape[1])); ym = dev.DeviceVector(M.shape[0])
M.mult(xm, ym)
t, _ = T(lambda: [M.mult(xm, ym) for _ in range(10)])
print("spmv M    %8.3f ms -> %.0f GB/s" % (t/10, (12*M.nnz + 8*M.shape[0]*2)/(t/10)/1e6))
