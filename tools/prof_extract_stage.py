"""Host-side profile of the "extract" stage of a bench workload (developer tool): python tools/prof_extract_stage.py [cfg3|cfg5|cfg4]"""
import cProfile, pstats, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if wl == "cfg3":
    os.environ["TIGAR_IMPLICIT_M"] = "1"
import numpy as np
import tigar_amd as t
from tigar_amd import BSplines as B, device as dev
import bench
d, p, nel = bench.WORKLOADS[wl] if hasattr(bench, "WORKLOADS") else {"cfg3": (3, 3, 256), "cfg4": (2, 4, 256), "cfg5": (2, 3, 128)}[wl]
nf, nl = (3, 2) if wl == "cfg5" else (1, 2 if wl == "cfg4" else 1)
cm = bench.quarter_annulus_mesh(p, nel) if wl == "cfg5" else B.ExplicitBSplineControlMesh([p] * d, [B.uniformKnots(p, -1.0 if wl == "cfg4" else 0., 1., nel) for _ in range(d)])
def build():
    gen = t.EqualOrderSpline(nf, cm)
    for f in range(nf):
        s0 = gen.getScalarSpline(f)
        if wl == "cfg5":
            gen.addZeroDofs(f, s0.getSideDofs(0, 0, nLayers=nl))
            continue
        for direction in range(d):
            for side in (0, 1):
                gen.addZeroDofs(f, s0.getSideDofs(direction, side, nLayers=nl))
    dev.sync()
    return gen
build(); build()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); g = build(); ts.append(time.perf_counter() - t0)
print("extract stage: %s ms" % [round(1e3 * x, 2) for x in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5): build()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
