"""One rank of the multi-rank -m gpu tests (launched by tigar_amd.launch.spawn_local from
tests/test_gpu_multirank.py): the hot path through the public API on a patch split into z-slabs,
rank-local results written to ``outdir/rank<r>.npz`` for comparison with the single-rank run."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dev_csr(A):
    from tigar_amd.device import DeviceCSR
    return DeviceCSR.from_scipy(A)


def main():
    outdir, d, p, nel, method = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import tigar_amd as t
    from tigar_amd import common as tc, BSplines as B, forms as F
    comm = tc.worldcomm
    dcomm = comm.device()
    rank_r, world_r, kind = dcomm.info()
    periodic0 = os.environ.get("TIGAR_TEST_PERIODIC0") == "1"      # direction 0 periodic (the slab direction stays open)
    explicit = os.environ.get("TIGAR_TEST_EXPLICIT_A")             # "device" / "scipy": an assembled A instead of a form
    # (optional) element counts per direction and the set of periodic directions (never the last one: the slab direction)
    nels = [int(v) for v in os.environ["TIGAR_TEST_NELS"].split(",")] if os.environ.get("TIGAR_TEST_NELS") else [nel] * d
    per = set(int(c) for c in os.environ.get("TIGAR_TEST_PERIODIC", "0" if periodic0 else ""))
    kv = [B.uniformKnots(p, 0., 1., nels[k], k in per) for k in range(d)]
    gen = t.EqualOrderSpline(comm, 1, B.ExplicitBSplineControlMesh([p] * d, kv))
    assert getattr(gen.M, "is_implicit", False)          # several ranks: no rank holds all rows of M
    sp0 = gen.getScalarSpline(0)
    for direction in range(d):
        if direction not in per:
            for side in (0, 1):
                gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    spline = t.ExtractedSpline(gen, 2 * p)
    from tigar_amd import device as dev
    dev.prof_reset()
    if explicit:
        # every rank holds the whole assembled matrix (with a coupling added by hand, as reef-knot.py:460-467 does)
        A = F.LaplaceForm().assemble_matrix(spline.V).to_scipy().tolil()
        A[5, A.shape[1] - 7] = 0.25
        A = A.tocsr()
        K = spline.extractMatrix(A if explicit == "scipy" else dev_csr(A), diag=1.5)
    else:
        K = spline.assembleMatrix(F.LaplaceForm(), diag=1.5)
    tensor_walks = dev.prof_get(5)[1]                    # final stages of the tensor-pattern PtAP that delivered rows of K
    f1 = lambda x: np.sin(np.pi * x)
    rhs = spline.assembleVector(F.SeparableLoadForm([f1] * d, scale=d * np.pi ** 2))
    solver = t.PETScKrylovSolver(*(method.split(":") if ":" in method else (method, "jacobi")))
    solver.parameters["relative_tolerance"] = 1e-10
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V, spline.localFERange())
    dev.prof_reset()
    U = spline.solveLinearSystem(K, rhs, u)
    its1 = solver.last["iterations"]
    overlapped = dev.prof_get(1)[1]                      # products that ran beside their halo exchange
    symgrid = dev.prof_get(7)[1]                         # CG solves on the half-storage copy of this rank's z slab
    host_waits = dev.prof_get(4)[1]                      # host waits of the communicator inside the exchanges of the solve
    resnorm = solver.last.get("residual_norm", np.nan)
    # second solve from the converged state: must stop at once (non-zero initial guess path with halo)
    solver.parameters["nonzero_initial_guess"] = True
    from tigar_amd.device import DeviceVector
    U2 = DeviceVector(data=U.get_local())
    its2 = solver.solve(K, U2, rhs)
    # the guess solveLinearSystem reads from u when the solver asks for one (the reference seeds MTU = M^T u.vector(),
    # tIGAr/common.py:1250-1254): rank-local FE rows + ghost rows of the neighbours, through the slab engine
    guess = spline._initial_guess_through_slabs(u).get_local()
    # optional: the same system at a ladder of tolerances (so that some solve ends with its norm just below the
    # tolerance, the case in which an iteration enqueued past convergence could come back to life)
    ladder_U, ladder_res, ladder_its = [], [], []
    if os.environ.get("TIGAR_TEST_RTOLS"):
        solver.parameters["nonzero_initial_guess"] = False
        for rt in [float(v) for v in os.environ["TIGAR_TEST_RTOLS"].split(",")]:
            solver.parameters["relative_tolerance"] = rt
            Ux = DeviceVector(U.size())
            ladder_its.append(solver.solve(K, Ux, rhs))
            ladder_U.append(Ux.get_local())
            ladder_res.append(solver.last["residual_norm"])
    Ks = K.to_scipy()
    g0, g1 = spline.localDofRange()
    r0, r1 = spline.localFERange()
    cp0 = gen.cpFuncs[0].vector().get_local()
    np.savez(os.path.join(outdir, "rank%d.npz" % comm.rank), g=np.array([g0, g1, r0, r1]),
             K_indptr=Ks.indptr, K_indices=Ks.indices, K_data=Ks.data, rhs=rhs.get_local(), U=U.get_local(),
             u=u.vector().get_local(), its=np.array([its1, its2]), guess=guess,
             comm=np.array([rank_r, world_r, dev.Comm.KINDS.index(kind)]), cp0=cp0,
             U2=U2.get_local(), overlapped=np.array([overlapped]), host_waits=np.array([host_waits]),
             resnorm=np.array([resnorm]), ladder_U=np.array(ladder_U), ladder_res=np.array(ladder_res),
             ladder_its=np.array(ladder_its), tensor_walks=np.array([tensor_walks]), symgrid=np.array([symgrid]))
    comm.barrier()


if __name__ == "__main__":
    main()
