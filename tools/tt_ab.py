"""A/B of the tensor-pattern PtAP passes inside ONE process (rates differ between processes by buffer placement):
x+y passes over a block of FE planes, variants selected by environment variables read per launch.
usage: tt_ab.py p nel planes rounds VAR     (VAR: an environment switch the library reads per launch)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm
from tigar_amd.kronptap import KronExtraction
from tigar_amd.tensorptap import TensorPtAP
p, nel, planes, rounds, var = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
d = 3
basis = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0., 1., nel)] * d).getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
plan = TensorPtAP.for_extraction(KronExtraction(basis, grid))
n1 = nel * p + 1
z0 = 3 * p
z1 = z0 + planes
A = LaplaceForm().assemble_matrix(V, z0 * n1 * n1, z1 * n1 * n1)
dev.sync()
res = {}
for r in range(rounds):
    for name in ("default", var):
        os.environ.pop(var, None)
        if name == var:
            os.environ[var] = "1"
        dev.sync(); dev.timer_start(0)
        piece = plan.planes(A, z0 * n1 * n1, z0, z1)
        t_xy = dev.timer_stop(0)
        assert piece is not None
        res.setdefault(name, []).append(t_xy)
        del piece
for name, v in res.items():
    v = np.array(v[1:])
    print("%-20s x+y %.3f ms (min %.3f max %.3f)" % (name, v.mean(), v.min(), v.max()))
