"""developer tool: the mapped FE assembly alone (pattern kernel + element kernel) on a rational volume map.

    python tools/asm_bench.py [p] [nel] [form] [reps]          form: laplace | mass | elast<i><j> (block of the elasticity form)
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402
from tigar_amd import device as dev              # noqa: E402
from tigar_amd import common as tc               # noqa: E402


def main():
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nel = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    form = sys.argv[3] if len(sys.argv) > 3 else "laplace"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dev.device_info()
    gen = tc.EqualOrderSpline(tc.selfcomm, 1, bench.rational_volume_mesh(p, nel))
    g = gen.V.grids[0]
    uks = [np.asarray(g.vertices[k]) for k in range(3)]
    cp = [f.vector() for f in gen.cpFuncs]
    nelem = nel ** 3
    for r in range(reps + 1):
        dev.sync()
        t0 = time.perf_counter()
        if form.startswith("elast"):
            A = dev.assemble_mapped_elasticity_block(uks, p, cp, int(form[5]), int(form[6]), 1.3, 0.7)
        else:
            A = dev.assemble_mapped_matrix(uks, p, cp, form)
        dev.sync()
        dt = time.perf_counter() - t0
        nnz = A.nnz
        del A
        if r:
            fl = {"laplace": {1: 0, 2: 0, 3: 2 * 184e3}.get(p, 0), "mass": 0}.get(form, 0)
            print("p=%d nel=%d %s: %.3f ms  (%.1f ns/element, nnz %.3e, %.0f GB/s of 12 B/nnz%s)"
                  % (p, nel, form, dt * 1e3, dt / nelem * 1e9, nnz, 12.0 * nnz / dt / 1e9,
                     ", %.1f TFLOP/s" % (fl * nelem / dt / 1e12) if fl else ""))


if __name__ == "__main__":
    main()
