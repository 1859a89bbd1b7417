"""developer tool: the half-storage product on a grid of 259 x 259 x (nz + 3) points (the z slab of one of several ranks at cfg3) for
different numbers of z chunks (TIGAR_SYMGRID_CHUNKS; 0 = the library's choice).  usage: symgrid_slab_chunks.py [nz]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, device as dev
p = 3
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kv = [B.uniformKnots(p, 0., 1., 256), B.uniformKnots(p, 0., 1., 256), B.uniformKnots(p, 0., 1., nz)]
gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
s0 = gen.getScalarSpline(0)
for direction in range(3):
    for side in (0, 1):
        gen.addZeroDofs(0, s0.getSideDofs(direction, side))
spline = t.ExtractedSpline(gen, 2 * p)
K = spline.assembleMatrix(F.LaplaceForm())
rhs = spline.assembleVector(F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * 3, scale=3 * np.pi ** 2))
n = K.shape[0]
print("rows", n, "planes", nz + p, flush=True)
for ch in (0, 2, 3, 4, 5, 6, 8):
    os.environ["TIGAR_SYMGRID_CHUNKS"] = str(ch)
    os.environ["TIGAR_SPMV_SYM"] = "1"
    ks = t.PETScKrylovSolver("cg", "jacobi")
    ks.parameters["relative_tolerance"] = 1e-6
    best = None
    for rep in range(2):
        U = dev.DeviceVector(n)
        dev.prof_reset(); dev.sync()
        its = ks.solve(K, U, rhs)
        dev.sync()
        ms, cnt = dev.prof_get(0)
        best = ms / max(cnt, 1) if best is None else min(best, ms / max(cnt, 1))
    print("chunks %d: product %.4f ms (%d its)" % (ch, best, its), flush=True)
