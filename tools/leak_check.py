"""the path in a loop: free HBM + idle pool bytes must not drift (developer tool)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, device as dev
p, nel, d = 2, 96, 3
f = lambda x: np.sin(np.pi * x)
for it in range(25):
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, [B.uniformKnots(p, 0., 1., nel) for _ in range(d)]))
    s0 = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    spline.setSolverOptions(linearSolver=t.PETScKrylovSolver("cg", "jacobi"))
    u = t.Function(spline.V)
    spline.solveLinearVariationalProblem(F.Equation(F.LaplaceForm(), F.SeparableLoadForm([f] * d, scale=d * np.pi ** 2)), u)
    del gen, spline, u
    dev.sync()
    if it % 6 == 0 or it == 24:
        free_b, tot = dev.mem_info()
        pooled, nblk, live = dev.pool_stats()
        print("pass %2d: free %.3f GB + pooled %.3f GB = %.3f GB, pool blocks %d, live blocks %d" %
              (it, free_b / 1e9, pooled / 1e9, (free_b + pooled) / 1e9, nblk, live), flush=True)
