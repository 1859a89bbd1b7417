"""A/B of FE-input generator variants INSIDE one process (launch times differ by +-15 % between processes and boxes):
usage gen_ab.py p nel planes rounds  ->  average ms per variant, variants interleaved round by round"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm
p, nel, planes, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = 3
basis = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0., 1., nel)] * d).getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
lap = LaplaceForm()
n1 = nel * p + 1
r0, r1 = 10 * n1 * n1, (10 + planes) * n1 * n1
variants = [("threads", {"TIGAR_KRON3_THREADS": "1"}), ("rows", {})]
acc = {k: [] for k, _ in variants}
A = lap.assemble_matrix(V, r0, r1); dev.sync(); nnz = A.nnz; del A
for r in range(rounds):
    for name, env in variants:
        for k in ("TIGAR_KRON3_THREADS", "TIGAR_K3_Z"):
            os.environ.pop(k, None)
        os.environ.update(env)
        dev.sync()
        dev.timer_start(0)
        A = lap.assemble_matrix(V, r0, r1)
        ms = dev.timer_stop(0)
        del A
        acc[name].append(ms)
for name, _ in variants:
    v = np.array(acc[name][1:])
    print("%-10s %.3f ms (min %.3f max %.3f) -> %.2f TB/s" % (name, v.mean(), v.min(), v.max(), 12 * nnz / v.mean() / 1e9))
