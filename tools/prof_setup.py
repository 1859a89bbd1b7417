"""host-side profile of the generator set-up at cfg3 (developer tool)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tigar_amd import device as dev
from tigar_amd.common import EqualOrderSpline, ExtractedSpline, selfcomm
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
p, nel, d = 3, 256, 3
def make():
    cm = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0., 1., nel)] * d)
    gen = EqualOrderSpline(selfcomm, 1, cm)
    sp0 = gen.getScalarSpline(0)
    for direction in range(d):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    dev.sync()
    return gen
make(); make()
t = time.perf_counter(); make(); print("set-up %.1f ms" % (1e3 * (time.perf_counter() - t)))
pr = cProfile.Profile(); pr.enable(); make(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
