"""One rank of the multi-rank Newton test (launched by tigar_amd.launch.spawn_local from tests/test_gpu_multirank.py):
solveNonlinearVariationalProblem on a patch split into z-slabs -- u, du and the IGA dofs rank-local, the norm of M^T R
global, the forms reading u through its ghosted form (tIGAr/common.py:1304-1348 on distributed vectors)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def problem(comm, d, p, nel):
    import tigar_amd as t
    from tigar_amd import BSplines as B
    from oracle import tigar_oracle as O
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    gen = t.EqualOrderSpline(comm, 1, B.ExplicitBSplineControlMesh([p] * d, kv))
    sp0 = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p, comm=comm)
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-13
    spline.setSolverOptions(maxIters=20, relativeTolerance=1e-9, linearSolver=solver)
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    X, _ = O.fe_node_grid(s)
    exact = np.prod(np.sin(np.pi * X), axis=1)
    f = d * np.pi ** 2 * exact + exact ** 3
    return gen, spline, f


def cube(v):
    return v.pointwise_mult(v).pointwise_mult(v)


def dcube(v):
    w = v.pointwise_mult(v)
    w.axpy(2.0, w.copy())
    return w


def run(comm, d, p, nel, with_dofs):
    import tigar_amd as t
    from tigar_amd import forms as F, device as dev
    gen, spline, f = problem(comm, d, p, nel)
    u = spline.localFunction()
    res = F.SemilinearResidual(u, f, cube, dcube)
    dofs = None
    if with_dofs:
        g0, g1 = spline.localDofRange()
        dofs = dev.DeviceVector(data=np.zeros(g1 - g0))
    hist = spline.solveNonlinearVariationalProblem(res, res.tangent(), u, igaDoFs=dofs)
    return spline, u, dofs, hist


def main():
    outdir, d, p, nel, with_dofs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1"
    from tigar_amd import common as tc
    comm = tc.worldcomm
    spline, u, dofs, hist = run(comm, d, p, nel, with_dofs)
    g0, g1 = spline.localDofRange()
    r0, r1 = spline.localFERange()
    full = spline.gatherFunction(u)                  # the replicated function again (every rank: all FE rows)
    assert full.local_range is None and full.vector().size() == spline.V.dim()
    np.savez(os.path.join(outdir, "rank%d.npz" % comm.rank), g=np.array([g0, g1, r0, r1]), hist=np.array(hist),
             u=u.vector().get_local(), dofs=dofs.get_local() if dofs is not None else np.zeros(0),
             u_gathered=full.vector().get_local())
    comm.barrier()


if __name__ == "__main__":
    main()
