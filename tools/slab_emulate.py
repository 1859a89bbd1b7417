"""Setup phase (FE input + PtAP + M^T b) of ONE rank of an N-GPU z-slab run, on a single GPU and without a
communicator (developer tool: the setup needs no communication, so the per-rank workload of a multi-GPU
job can be timed and profiled on one device).   usage: slab_emulate.py p nel rank world [sub_planes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tigar_amd import device as dev
from tigar_amd.common import TensorFunctionSpace
from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
from tigar_amd.forms import LaplaceForm, SeparableLoadForm
from tigar_amd.dist import SlabHotPath
from tigar_amd.dist import pick_sub_planes

p, nel, rank, world = (int(a) for a in sys.argv[1:5])
d = 3
basis = ExplicitBSplineControlMesh([p] * d, [uniformKnots(p, 0., 1., nel)] * d).getScalarSpline()
grid = basis.generateMesh(degree=p)
V = TensorFunctionSpace([grid], "Lagrange")
lap = LaplaceForm()
load = SeparableLoadForm([lambda x: np.sin(np.pi * x)] * d, scale=d * np.pi ** 2)
zd = []
for direction in range(d):
    for side in (0, 1):
        zd += basis.getSideDofs(direction, side)
zd = np.asarray(zd, dtype=np.int32)
free_b, _ = dev.mem_info()
probe = SlabHotPath(basis, grid, rank, world, None)
planes = probe.k1 - probe.k0
sub = int(sys.argv[5]) if len(sys.argv) > 5 else pick_sub_planes(d, p, nel, planes, free_b)
path = SlabHotPath(basis, grid, rank, world, None, sub_planes=sub)
for it in range(int(os.environ.get("PASSES", "2"))):
    timers = {}
    t0 = time.perf_counter()
    K, rhs = path.assemble(lambda r0, r1: lap.assemble_matrix(V, r0, r1), lambda r0, r1: load.assemble_vector(V, r0, r1),
                           zd, 1.0, timers)
    dev.sync()
    print("rank %d/%d: %d planes in sub-slabs of %d, pass %d: %.3f s" % (rank, world, planes, sub, it, time.perf_counter() - t0),
          {k: round(v, 4) for k, v in timers.items()}, "nnz(K_loc) =", K.nnz, flush=True)
    if it == int(os.environ.get("PASSES", "2")) - 1:
        # the product of the Krylov solve on this rank's row block (x addressed by global column)
        x = dev.DeviceVector(data=np.random.default_rng(0).standard_normal(K.shape[1]))
        y = dev.DeviceVector(K.shape[0])
        for label, en in (("CSR", False), ("sliced copy", True)):
            info = K.spmv_sell(en)
            K.mult(x, y); dev.sync()
            dev.timer_start(0)
            for _ in range(20):
                K.mult(x, y)
            ms = dev.timer_stop(0) / 20
            print("  K_loc x on %s %s: %.3f ms" % (label, info, ms), flush=True)
        K.spmv_sell(False)
    del K, rhs
