"""One rank of the multi-rank mapped-geometry test (launched by tigar_amd.launch.spawn_local from tests/test_gpu_assembly.py):
Poisson on a rational volume map (or, kind "identity", the identity map through the SAME mapped forms), the patch split
into z-slabs -- every rank evaluates the control functions on the window of FE planes its row blocks need and assembles its
rows of the FE matrix / vector there (dolfin.assemble on a distributed mesh, tIGAr/common.py:1206-1220)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load(x):
    return np.sin(x[:, 0]) + x[:, 1] * x[:, 2]


def problem(comm, kind, p, nels):
    import tigar_amd as t
    from tigar_amd import BSplines as B, NURBS as N
    if kind == "volume":
        from geom_util import rational_volume
        kvs, C = rational_volume(p, nels)
        cm = N.NURBSControlMesh([p] * 3, kvs, C)
    else:
        cm = B.ExplicitBSplineControlMesh([p] * 3, [B.uniformKnots(p, 0., 1. + 0.5 * k, nels[k]) for k in range(3)])
    gen = t.EqualOrderSpline(comm, 1, cm)
    sp0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p, comm=comm)
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-11
    spline.setSolverOptions(linearSolver=solver)
    return gen, spline, solver


def run(comm, kind, p, nels):
    import tigar_amd as t
    from tigar_amd import forms as F
    gen, spline, solver = problem(comm, kind, p, nels)
    K = spline.assembleMatrix(F.LaplaceForm(geometry=gen), diag=1.5)
    rhs = spline.assembleVector(F.NodalLoadForm(load, gen))
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(K, rhs, u)
    return gen, spline, K, rhs, U, u, solver.last["iterations"]


def main():
    outdir, kind, p = sys.argv[1], sys.argv[2], int(sys.argv[3])
    nels = [int(v) for v in sys.argv[4].split(",")]
    from tigar_amd import common as tc
    comm = tc.worldcomm
    gen, spline, K, rhs, U, u, its = run(comm, kind, p, nels)
    Ks = K.to_scipy()
    g0, g1 = spline.localDofRange()
    r0, r1 = spline.localFERange()
    np.savez(os.path.join(outdir, "rank%d.npz" % comm.rank), g=np.array([g0, g1, r0, r1]), K_data=Ks.data,
             K_indices=Ks.indices, K_indptr=Ks.indptr, rhs=rhs.get_local(), U=U.get_local(), u=u.vector().get_local(),
             its=np.array([its]))
    comm.barrier()


if __name__ == "__main__":
    main()
