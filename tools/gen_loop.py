"""FE input generator in a loop (developer tool for overlap experiments): usage gen_loop.py p nel planes seconds"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm
p, nel, planes, secs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
d = 3
basis = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0., 1., nel)] * d).getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
lap = LaplaceForm()
n1 = nel * p + 1
r0, r1 = 10 * n1 * n1, (10 + planes) * n1 * n1
A = lap.assemble_matrix(V, r0, r1); dev.sync(); nnz = A.nnz; del A
t_end = time.time() + secs
n = 0
t0 = time.perf_counter()
while time.time() < t_end:
    A = lap.assemble_matrix(V, r0, r1)
    dev.sync()
    del A
    n += 1
dt = time.perf_counter() - t0
print("generator: %d x %.1f GB in %.2f s -> %.2f TB/s" % (n, 12 * nnz / 1e9, dt, n * 12 * nnz / dt / 1e12), flush=True)
