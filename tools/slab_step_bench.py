"""developer tool: the stages of one step on a patch of 256 x 256 x nz cubic elements on ONE GPU -- what one of several ranks
computes on its z slab of cfg3 (without the exchanges): 32 layers 96 ms (PtAP 22, solve 70.5 = 100 iterations), 64: 161 ms,
128: 292 ms, 256 (cfg3 itself): 591 ms.  usage: slab_step_bench.py [nz]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TIGAR_IMPLICIT_M", "1")
import tigar_amd as t
from tigar_amd import BSplines as B, forms as F, device as dev
from tigar_amd.common import Function
p = 3
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kv = [B.uniformKnots(p, 0., 1., 256), B.uniformKnots(p, 0., 1., 256), B.uniformKnots(p, 0., 1., nz)]
cm = B.ExplicitBSplineControlMesh([p] * 3, kv)
lap = F.LaplaceForm()
load = F.SeparableLoadForm([lambda x: np.sin(np.pi * x)] * 3, scale=3 * np.pi ** 2)
for rep in range(4):
    ts = [time.perf_counter()]
    def mark():
        dev.sync(); ts.append(time.perf_counter())
    gen = t.EqualOrderSpline(1, cm)
    s0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
    mark()
    spline = t.ExtractedSpline(gen, 2 * p)
    K = spline.assembleMatrix(lap)
    mark()
    rhs = spline.assembleVector(load)
    mark()
    ks = t.PETScKrylovSolver("cg", "jacobi")
    ks.parameters["relative_tolerance"] = 1e-6
    spline.setSolverOptions(linearSolver=ks)
    u = Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    mark()
    d = np.diff(ts)
    print("nz %d: dofs %d: extract %.1f  ptap %.1f  mtb %.1f  solve %.1f (its %d)  total %.1f ms" % (nz, K.shape[0], d[0]*1e3, d[1]*1e3, d[2]*1e3, d[3]*1e3, ks.last["iterations"], 1e3*(ts[-1]-ts[0])), flush=True)
    del K, U, u, rhs, spline, gen
