"""Random patches through the Newton driver (solveNonlinearVariationalProblem, tIGAr/common.py:1304-1348) on
-lap u + u^3 = f against ``oracle.newton_semilinear`` (developer tool): the history of relative norms and the solution.

    python tests/fuzz/fuzz_newton.py [cases]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import tigar_oracle as O  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def main():
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = 0
    for i in range(N):
        rng = np.random.default_rng([123, i])
        while True:
            case = fz.draw_case(rng, 6000)
            if case["nfields"] == 1 and case["d"] <= 2:
                break
        case["bc"] = "sides"
        try:
            kvs = fz.knot_vectors(case, B.uniformKnots)
            gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh(case["ps"], kvs))
            sp0 = gen.getScalarSpline(0)
            so = O.BSpline(case["ps"], kvs)
            for k in range(case["d"]):
                if case["kinds"][k] != "periodic":
                    for side in (0, 1):
                        gen.addZeroDofs(0, sp0.getSideDofs(k, side))
            if all(kd == "periodic" for kd in case["kinds"]):
                gen.addZeroDofs(0, [0])
            if len(set(gen.zeroDofs)) >= so.getNcp():
                continue            # every dof constrained: the initial residual is 0 and the reference divides by it (:1333)
            spline = t.ExtractedSpline(gen, 2 * max(case["ps"]))
            solver = t.PETScKrylovSolver("cg", "jacobi")
            solver.parameters["relative_tolerance"] = 1e-13
            solver.parameters["maximum_iterations"] = 50000
            spline.setSolverOptions(maxIters=25, relativeTolerance=1e-9, linearSolver=solver)
            X, _ = O.fe_node_grid(so)
            exact = np.prod(np.sin(np.pi * X), axis=1)
            f = case["d"] * np.pi ** 2 * exact + exact ** 3
            u = t.Function(spline.V)
            cube = lambda v: v.pointwise_mult(v).pointwise_mult(v)

            def dcube(v):
                w = v.pointwise_mult(v)
                w.axpy(2.0, w.copy())
                return w
            res = F.SemilinearResidual(u, f, cube, dcube)
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                hist = spline.solveNonlinearVariationalProblem(res, res.tangent(), u)
            Mo = O.generate_M_tensor(so)
            Kfe = F.LaplaceForm().assemble_matrix(spline.V).to_scipy()
            Mfe = F.MassForm().assemble_matrix(spline.V).to_scipy()
            uo, Uo, ho = O.newton_semilinear(Mo, Kfe, Mfe, f, lambda v: v ** 3, lambda v: 3 * v ** 2, list(spline.zeroDofs), rtol=1e-9)
            assert len(hist) == len(ho), "iterations %d vs %d" % (len(hist), len(ho))
            for a, b in zip(hist, ho):
                assert abs(a - b) <= 1e-5 * max(b, 1e-12) + 1e-11, "history %s vs %s" % (hist, ho)
            e = np.max(np.abs(u.vector().get_local() - uo)) / max(1e-300, np.max(np.abs(uo)))
            assert e <= 1e-8, "solution %g" % e
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print("FAIL", json.dumps(case), type(ex).__name__, str(ex)[:300], flush=True)
    print(json.dumps({"cases": N, "failed": bad}))


if __name__ == "__main__":
    main()
