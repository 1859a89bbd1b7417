"""developer tool / profile line: the hot path on a div-conforming B-spline space (BSplineCompat "RT": the space of
demos/taylor-green/taylor-green-3d.py:42-90) with the demo's solver setting (GMRES + Jacobi, rtol 1e-2).

    python tools/rt_bench.py [nel] [base degree] [steps]      (env TIGAR_IMPLICIT_M=1: streamed in sub-slabs)

One step = generator (M per field, control functions) -> ExtractedSpline -> M^T A M of the linear-elasticity form on the
common Q_(k+1) grid (nine blocks, each by the line walks with different row / column bases, the Kronecker-sum blocks formed
inside the first pass when streamed) -> M^T b -> GMRES(30) + Jacobi -> u = M U.  Prints one JSON line."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nel = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F, device as dev, common as tc
    from tigar_amd.compatibleSplines import BSplineCompat
    info = dev.device_info()
    degs = [k] * 3
    kv = [B.uniformKnots(k, 0.0, np.pi, nel) for _ in range(3)]
    cm = B.ExplicitBSplineControlMesh(degs, kv)
    form = F.ElasticityForm(2.0, 1.0)
    stages, last = {}, {}

    def step(record):
        ts = [time.perf_counter()]
        rec = {}

        def mark(name):
            dev.sync()
            ts.append(time.perf_counter())
            rec[name] = ts[-1] - ts[-2]
        gen = BSplineCompat(tc.selfcomm, cm, "RT", degs)
        for f in range(3):
            s0 = gen.getFieldSpline(f)
            for side in (0, 1):
                gen.addZeroDofs(f, s0.getSideDofs(f, side))
        mark("extract")
        spline = t.ExtractedSpline(gen, 2 * (k + 1))
        mark("transpose")
        if getattr(gen.M, "is_implicit", False):
            K = spline.assembleMatrix(form)
        else:
            if "A" not in last:
                last["A"] = form.assemble_matrix(spline.V)       # (FE input resident before the timed region, as cfg2)
                dev.sync()
                ts[-1] = time.perf_counter()
            K = spline.extractMatrix(last["A"])
        mark("ptap")
        if "b" not in last:
            rng = np.random.default_rng(3)
            last["b"] = dev.DeviceVector(data=rng.standard_normal(spline.V.dim()))
            dev.sync()
            ts[-1] = time.perf_counter()
        rhs = spline.extractVector(last["b"])
        mark("mtb")
        solver = t.PETScKrylovSolver("gmres", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-2
        spline.setSolverOptions(linearSolver=solver)
        u = t.Function(spline.V, spline.localFERange() if getattr(gen.M, "is_implicit", False) else None)
        spline.solveLinearSystem(K, rhs, u)
        mark("solve")
        if record:
            for kk, v in rec.items():
                stages.setdefault(kk, []).append(v)
        last.update(ncp=K.shape[0], nnzK=K.nnz, its=solver.last["iterations"], implicit=bool(getattr(gen.M, "is_implicit", False)),
                    fields=[[s1.p for s1 in gen.getFieldSpline(f).splines] for f in range(3)], fe=spline.V.dim())
        del K, spline, gen

    step(False)
    dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    dev.sync()
    el = (time.perf_counter() - t0) / steps
    print(json.dumps({"metric": "DoF/s (extraction + M^T A M + M^T b + GMRES solve + prolongation)", "value": last["ncp"] / el,
                      "unit": "DoF/s", "ms_per_step": 1e3 * el, "steps": steps, "dtype": "f64",
                      "config": {"workload": "BSplineCompat RT, base degree %d, %d^3 elements, linear-elasticity form, GMRES(30)+Jacobi "
                                             "rtol 1e-2 (demos/taylor-green/taylor-green-3d.py:89-91)" % (k, nel),
                                 "field_degrees": last["fields"], "dofs": last["ncp"], "fe_rows": last["fe"], "nnz_K": last["nnzK"],
                                 "gmres_iterations": last["its"], "M_implicit": last["implicit"],
                                 "stages_s": {kk: round(float(np.mean(v)), 6) for kk, v in stages.items()},
                                 "tensor_walk_final_passes": dev.prof_get(5)[1], "device": info}}))


if __name__ == "__main__":
    main()
