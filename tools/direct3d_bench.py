"""The default solver (``linearSolver=None``: the reference's direct solve, tIGAr/common.py:1255-1256) on a 3-D Poisson system
beyond the banded LU's budget: the banded Cholesky factorisation of csrc/tg_chol.hip.  usage: direct3d_bench.py [p] [nel]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, common as tc, device as dev
p = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nel = int(sys.argv[2]) if len(sys.argv) > 2 else 48
kv = [B.uniformKnots(p, 0., 1., nel)] * 3
gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
sp0 = gen.getScalarSpline(0)
for direction in range(3):
    for side in (0, 1):
        gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
spl = t.ExtractedSpline(gen, 2 * p)
f1 = lambda x_: np.sin(np.pi * x_)
K, b = spl.assembleLinearSystem(F.LaplaceForm(), F.SeparableLoadForm([f1] * 3, scale=3 * np.pi ** 2))
kl, ku, nb = dev.lu_band_info(K)
print("K: %d dofs, %d entries, half-bandwidth %d; LU band %.1f GB, Cholesky band %.1f GB, n kl^2 = %.2e"
      % (K.shape[0], K.nnz, kl, nb / 1e9, 8e-9 * K.shape[0] * (kl + 1), float(K.shape[0]) * kl * kl), flush=True)
for rep in range(2):
    d = tc._default_linear_solver()
    x = dev.DeviceVector(K.shape[0])
    dev.sync(); t0 = time.perf_counter()
    d.solve(K, x, b)
    dev.sync(); dt = time.perf_counter() - t0
    r = K.mult(x); r.axpy(-1.0, b)
    print("default solver: %s, %.3f s, relative residual %.2e" % ({k: d.last[k] for k in ("solver", "factorisation") if k in d.last}, dt, r.norm() / b.norm()), flush=True)
