"""SpMV micro-benchmark on a K = M^T A M of a 3-D patch (developer tool)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm
from tigar_amd.dist import SlabHotPath
p, nel = int(sys.argv[1]), int(sys.argv[2])
d = 3
basis = ExplicitBSplineControlMesh([p]*d, [uniformKnots(p, 0., 1., nel)]*d).getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
lap = LaplaceForm()
path = SlabHotPath(basis, grid, sub_planes=int(sys.argv[3]) if len(sys.argv) > 3 else None)
K, rhs = path.assemble(lambda a, b: lap.assemble_matrix(V, a, b), lambda a, b: dev.DeviceVector(b - a), [])
n = K.shape[0]
x = dev.DeviceVector(data=np.random.default_rng(0).standard_normal(n)); y = dev.DeviceVector(n)
if os.environ.get('SELL', '1') != '0':
    print('sliced copy:', K.spmv_sell(True))
K.mult(x, y); dev.sync()
reps = 20
dev.timer_start(0)
for _ in range(reps): K.mult(x, y)
ms = dev.timer_stop(0) / reps
b = 12*K.nnz + 4*(n+1) + 16*n
print("p=%d nel=%d nnzK=%d  spmv %.3f ms -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  env CAP=%s NT=%s" % (p, nel, K.nnz, ms, b/ms/1e6, b/ms/1e6/80, os.environ.get("TIGAR_SPMV_CAP"), os.environ.get("TIGAR_SPMV_NT")))
