import cProfile, pstats, time, os, sys
sys.path.insert(0, os.getcwd())
os.environ["TIGAR_IMPLICIT_M"] = "1"
import numpy as np
import tigar_amd as t
from tigar_amd import BSplines as B, device as dev
p, nel, d = 3, 256, 3
kv = [B.uniformKnots(p, 0., 1., nel) for _ in range(d)]
cm = B.ExplicitBSplineControlMesh([p] * d, kv)
def build():
    gen = t.EqualOrderSpline(1, cm)
    s0 = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, s0.getSideDofs(direction, side))
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
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
