"""One rank of the multi-field multi-rank -m gpu tests: several fields on one tensor basis (EqualOrderSpline(nF > 1))
split into z-slabs -- rows of K, M^T b, the Krylov solution and the prolongation of this rank, with the reference
(field-after-field) index of every local dof, written to ``outdir/rank<r>.npz``."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hashed(n, seed):
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def problem(case, comm):
    """(generator, spline, K, rhs, solver method) of a test case, through the public API on communicator ``comm``"""
    import scipy.sparse as sp
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    if case == "rt3d":
        # the space of demos/taylor-green/taylor-green-3d.py:42-50: BSplineCompat("RT") -- three fields on DIFFERENT bases (one
        # degree higher along their own direction) over one Q_(k+1) node grid, normal-direction boundary conditions
        from tigar_amd.compatibleSplines import BSplineCompat
        degs = [int(v) for v in os.environ.get("TIGAR_TEST_FDEGS", "1,1,1").split(",")]
        nels = [int(v) for v in os.environ.get("TIGAR_TEST_FNELS", "5,4,9").split(",")]
        kv = [B.uniformKnots(degs[k], 0., 1. + 0.25 * k, nels[k]) for k in range(3)]
        gen = BSplineCompat(comm, B.ExplicitBSplineControlMesh(degs, kv), "RT", degs)
        for f in range(3):
            s0 = gen.getFieldSpline(f)
            for side in (0, 1):
                gen.addZeroDofs(f, s0.getSideDofs(f, side))
        spline = t.ExtractedSpline(gen, 2 * (max(degs) + 1))
        K = spline.assembleMatrix(F.ElasticityForm(2.0, 1.0), diag=1.5)
        rhs = spline.extractVector(hashed(spline.V.dim(), 79))
        return gen, spline, K, rhs, "gmres"
    if case == "mapped_elasticity3d":
        # three displacement fields on a rational volume (a NURBS control mesh): ElasticityForm(geometry=...) hands out row
        # blocks of its nine field blocks from the element kernels, on the control-function window of the rank's slab
        from tigar_amd import NURBS
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from geom_util import rational_volume
        p = int(os.environ.get("TIGAR_TEST_FP", 2))
        nels = [int(v) for v in os.environ["TIGAR_TEST_FNELS"].split(",")] if os.environ.get("TIGAR_TEST_FNELS") else [4, 3, 7]
        kvs, C = rational_volume(p, nels)
        gen = t.EqualOrderSpline(comm, 3, NURBS.NURBSControlMesh([p] * 3, kvs, C))
        for f in range(3):
            gen.addZeroDofs(f, gen.getScalarSpline(f).getSideDofs(2, 0))
        spline = t.ExtractedSpline(gen, 2 * p)
        K = spline.assembleMatrix(F.ElasticityForm(2.0, 1.0, geometry=gen), diag=1.5)
        rhs = spline.extractVector(hashed(spline.V.dim(), 80))
        return gen, spline, K, rhs, "cg"
    if case == "shell2d":            # cfg5-like: 2-D p=3, three fields, hashed non-symmetric A on the 3-field pattern
        d, p, nel, nF, method = 2, 3, 14, 3, "gmres"
    else:                            # 3-D elasticity, p=2, three fields (ElasticityForm: blocks as Kronecker sums)
        d, p, nel, nF, method = 3, 2, 7, 3, "cg"
    # (optional overrides, set by the tests for both the single-rank reference and the ranks: degree, element counts per
    #  direction, periodic directions other than the last)
    p = int(os.environ.get("TIGAR_TEST_FP", p))
    nels = [int(v) for v in os.environ["TIGAR_TEST_FNELS"].split(",")] if os.environ.get("TIGAR_TEST_FNELS") else [nel] * d
    per = set(int(c) for c in os.environ.get("TIGAR_TEST_FPER", ""))
    kv = [B.uniformKnots(p, 0., 1., nels[k], k in per) for k in range(d)]
    gen = t.EqualOrderSpline(comm, nF, B.ExplicitBSplineControlMesh([p] * d, kv))
    for f in range(nF):
        s0 = gen.getScalarSpline(f)
        if case == "shell2d":
            gen.addZeroDofs(f, s0.getSideDofs(0, 0, nLayers=2))
        else:
            gen.addZeroDofs(f, s0.getSideDofs(d - 1, 0))
    spline = t.ExtractedSpline(gen, 2 * p)
    nfe = spline.V.dim() // nF
    if case == "shell2d":
        pat = F.LaplaceForm().assemble_matrix(t.TensorFunctionSpace([gen.getScalarSpline(0).generateMesh(degree=p)],
                                                                    "Lagrange")).to_scipy().tocsr()
        pat.sort_indices()
        blocks = [[None] * nF for _ in range(nF)]
        for a in range(nF):
            for b in range(nF):
                Bk = pat.copy()
                Bk.data = 0.05 * hashed(Bk.nnz, 3 * a + b)
                blocks[a][b] = Bk if a != b else (Bk + 4.0 * sp.identity(nfe, format="csr")).tocsr()
        A = sp.bmat(blocks, format="csr")
        K = spline.extractMatrix(A, diag=1.5)
        rhs = spline.extractVector(hashed(A.shape[0], 77))
    else:
        K = spline.assembleMatrix(F.ElasticityForm(2.0, 1.0), diag=1.5)
        rhs = spline.extractVector(hashed(spline.V.dim(), 78))
    return gen, spline, K, rhs, method


def main():
    outdir, case = sys.argv[1], sys.argv[2]
    import tigar_amd as t
    from tigar_amd import common as tc, device as dev
    comm = tc.worldcomm
    dcomm = comm.device()
    gen, spline, K, rhs, method = problem(case, comm)
    assert getattr(gen.M, "is_implicit", False) and gen.M.nfields == 3
    solver = t.PETScKrylovSolver(method, "jacobi")
    solver.parameters["relative_tolerance"] = 1e-10
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V, spline.localFERange())
    dev.prof_reset()
    U = spline.solveLinearSystem(K, rhs, u)
    Ks = K.to_scipy()
    np.savez(os.path.join(outdir, "rank%d.npz" % comm.rank), dofs=spline.localDofIndices(),
             new_of_old=spline._slab_path().new_of_old(), g=np.array(spline.localDofRange()),
             fe=np.array(spline.localFERange()), K_indptr=Ks.indptr, K_indices=Ks.indices, K_data=Ks.data,
             rhs=rhs.get_local(), U=U.get_local(), u=u.vector().get_local(), its=np.array([solver.last["iterations"]]),
             host_waits=np.array([dev.prof_get(4)[1]]), kind=np.array([dev.Comm.KINDS.index(dcomm.info()[2])]))
    comm.barrier()


if __name__ == "__main__":
    main()
