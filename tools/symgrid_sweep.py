"""Sweep of TIGAR_SYMGRID_CHUNKS (z chunks per patch) for the half-storage product at the benchmark size:
python tools/symgrid_sweep.py [nel] [p] chunks..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nel, p = int(sys.argv[1]), int(sys.argv[2])
    chunks = [int(c) for c in sys.argv[3:]]
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev
    kv = [B.uniformKnots(p, 0., 1., nel)] * 3
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kv))
    s0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    K = spline.assembleMatrix(F.LaplaceForm())
    rhs = spline.assembleVector(F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * 3, scale=3 * np.pi ** 2))
    n = K.shape[0]
    for c in chunks:
        os.environ["TIGAR_SYMGRID_CHUNKS"] = str(c)
        ks = t.PETScKrylovSolver("cg", "jacobi")
        ks.parameters["relative_tolerance"] = 1e-6
        best = 1e9
        for rep in range(2):
            U = dev.DeviceVector(n)
            dev.prof_reset()
            its = ks.solve(K, U, rhs)
            dev.sync()
            ms, cnt = dev.prof_get(0)
            best = min(best, ms / max(cnt, 1))
        print("chunks %3d: product %.3f ms (%d its)" % (c, best, its), flush=True)


main()
