"""Random box-stencil matrices through the half-storage product (csrc/tg_symgrid.hip) against scipy (developer tool): stencil
radius 1-4, grid sizes that the 24 x 16 patches and the 64-row sub-steps never divide, random z cuts into slabs (the several-rank
form: entries above the slab gathered only, entries below it from the CSR rows), random numbers of z chunks, matrices that must
be declined (non-symmetric, an entry moved).

    python tests/fuzz/fuzz_symgrid.py [--seed S] [--cases N]"""
import argparse
import itertools
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def box_stencil(rng, shape, reach, symmetric=True):
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape[::-1])
    rows, cols = [], []
    for off in itertools.product(*[range(-reach, reach + 1)] * 3):
        src = [slice(max(0, -o), s - max(0, o)) for o, s in zip(off[::-1], shape[::-1])]
        dst = [slice(max(0, o), s - max(0, -o)) for o, s in zip(off[::-1], shape[::-1])]
        rows.append(idx[tuple(src)].ravel())
        cols.append(idx[tuple(dst)].ravel())
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, n))
    if symmetric:
        A = (A + A.T).tocsr()
    A.sort_indices()
    return A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=100)
    args = ap.parse_args()
    from tigar_amd import device as dev
    bad = 0
    for case in range(args.cases):
        rng = np.random.default_rng([args.seed, case])
        reach = int(rng.integers(1, 5))
        shape = [int(rng.integers(16, 70)), int(rng.integers(16, 50)), int(rng.integers(2 * reach + 2, 40))]
        while np.prod(shape) * (2 * reach + 1) ** 3 > 2.5e7:
            k = int(np.argmax(shape))
            shape[k] = max(16 if k < 2 else 2 * reach + 2, shape[k] - 5)
        shape = tuple(shape)
        os.environ["TIGAR_SYMGRID_CHUNKS"] = str(int(rng.integers(0, 9)))
        desc = {"case": case, "shape": shape, "reach": reach, "chunks": os.environ["TIGAR_SYMGRID_CHUNKS"]}
        try:
            A = box_stencil(rng, shape, reach)
            x = rng.standard_normal(A.shape[0])
            dx = dev.DeviceVector(data=x)
            ref, scale = A @ x, np.abs(A) @ np.abs(x)
            y, info = dev.DeviceCSR.from_scipy(A).mult_symgrid(dx)
            assert info is not None, "declined"
            assert np.max(np.abs(y.get_local() - ref) / scale) < 1e-14, "whole matrix"
            n01, n2 = shape[0] * shape[1], shape[2]
            cuts = sorted(set([0, n2] + [int(c) for c in rng.integers(1, n2, size=int(rng.integers(1, 4)))]))
            for z0, z1 in zip(cuts[:-1], cuts[1:]):
                B = A[z0 * n01:z1 * n01].tocsr()
                B.sort_indices()
                yb, ib = dev.DeviceCSR.from_scipy(B).mult_symgrid(dx, row0=z0 * n01)
                if z1 - z0 < 2 * reach + 2:
                    assert ib is None, "a thin slab was accepted"
                    continue
                assert ib is not None, "slab declined"
                assert np.max(np.abs(yb.get_local() - ref[z0 * n01:z1 * n01]) / scale[z0 * n01:z1 * n01]) < 1e-14, "slab %d..%d" % (z0, z1)
            # must be declined: not symmetric; one value of the lower triangle changed
            if case % 5 == 0:
                N = box_stencil(rng, shape, reach, symmetric=False)
                assert dev.DeviceCSR.from_scipy(N).mult_symgrid(dx)[1] is None, "non-symmetric accepted"
                C = A.copy()
                r = int(rng.integers(n01, A.shape[0]))
                C.data[C.indptr[r]] += 0.25
                assert dev.DeviceCSR.from_scipy(C).mult_symgrid(dx)[1] is None, "asymmetric value accepted"
            # several fields on the grid (round 6): nf x nf blocks, symmetric as a whole; every third case
            if case % 3 == 0 and reach <= 3:
                nf = int(rng.integers(2, 4))
                sh = tuple(min(s_, m_) for s_, m_ in zip(shape, (40, 30, 20)))
                blocks = [[box_stencil(rng, sh, reach, symmetric=False) for _ in range(nf)] for _ in range(nf)]
                M = sp.bmat(blocks, format="csr")
                M = (M + M.T).tocsr()
                M.sort_indices()
                xm = rng.standard_normal(M.shape[0])
                ym, im = dev.DeviceCSR.from_scipy(M).mult_symgrid(dev.DeviceVector(data=xm))
                assert im is not None, "%d fields declined" % nf
                assert np.max(np.abs(ym.get_local() - M @ xm) / (np.abs(M) @ np.abs(xm))) < 1e-14, "%d fields" % nf
                M.data[M.indptr[M.shape[0] // nf + 7] + 1] += 0.25            # one value of an off-diagonal pair changed
                assert dev.DeviceCSR.from_scipy(M).mult_symgrid(dev.DeviceVector(data=xm))[1] is None, "asymmetric pair accepted"
        except AssertionError as e:
            bad += 1
            desc["error"] = str(e)
            print(json.dumps(desc), flush=True)
    print(json.dumps({"cases": args.cases, "failed": bad, "seed": args.seed}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
