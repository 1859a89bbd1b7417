"""Random symmetric positive definite band systems through the direct solve behind the C-ABI (`tg_lu_solve`: blocked banded
Cholesky of csrc/tg_chol.hip, substitutions on one or on several workgroups) against LAPACK's banded Cholesky (developer tool;
`tests/test_gpu_fuzz.py` runs a seeded set): sizes that the blocks of 32 columns and the 64 x 64 tiles never divide, half-widths
from 1 to beyond the matrix, bands with holes, a random number of sweep workgroups, one to seven panels per trailing update, and every fifth case a matrix the
factorisation must hand on to the LU (a pivot that is not positive, a value that breaks the symmetry).

    python tests/fuzz/fuzz_direct.py [--seed S] [--cases N]

TIGAR_POOL_POISON=1 fills what the allocator hands out with NaNs (reads of never-written memory show up)."""
import argparse
import json
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def spd_band(rng, n, kl, holes):
    offs = [o for o in range(1, kl + 1) if o == kl or not holes or rng.random() < 0.5]
    vals = [rng.standard_normal(n - o) for o in offs]
    B = sp.diags(vals + vals, offs + [-o for o in offs], shape=(n, n), format="csr")
    d = np.asarray(abs(B).sum(axis=1)).ravel() * (1.05 + rng.random(n)) + 1e-3
    A = (B + sp.diags(d)).tocsr()
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
        kind = int(rng.integers(0, 4))
        if kind == 0:                                   # short and wide
            n = int(rng.integers(2, 400)); kl = int(rng.integers(1, n))
        elif kind == 1:                                 # long and narrow
            n = int(rng.integers(1000, 40000)); kl = int(rng.integers(1, 70))
        else:
            kl = int(rng.integers(20, 1300)); n = int(rng.integers(kl + 1, max(kl + 2, 3000000 // kl)))
        wgs = int(rng.choice([0, 0, 2, 3, 5, 8, 11, 16, 40, 64]))
        if wgs:
            os.environ["TIGAR_CHOL_SWEEP_WGS"] = str(wgs)
        else:
            os.environ.pop("TIGAR_CHOL_SWEEP_WGS", None)
        os.environ["TIGAR_CHOL_SWEEP"] = "0" if rng.random() < 0.15 else "1"
        os.environ["TIGAR_CHOL_GROUP"] = str(int(rng.choice([0, 1, 2, 3, 4, 7])))       # panels per trailing update (0: by the band)
        os.environ["TIGAR_CHOL_FUSED"] = "0" if rng.random() < 0.3 else "1"             # one launch per block / a panel and an update kernel
        os.environ["TIGAR_CHOL_LOOKAHEAD"] = "0" if rng.random() < 0.15 else "1"
        desc = {"case": case, "n": n, "kl": kl, "wgs": wgs, "fused": os.environ["TIGAR_CHOL_FUSED"], "ahead": os.environ["TIGAR_CHOL_LOOKAHEAD"], "sweep": os.environ["TIGAR_CHOL_SWEEP"], "group": os.environ["TIGAR_CHOL_GROUP"]}
        try:
            A = spd_band(rng, n, kl, holes=rng.random() < 0.3)
            nrhs_x = rng.standard_normal(n)
            b = A @ nrhs_x
            ab = np.zeros((kl + 1, n))
            for o in range(kl + 1):
                ab[o, :n - o] = A.diagonal(-o)
            ref = sla.solveh_banded(ab, b, lower=True)
            c0 = dev.prof_get(8)[1]
            x = dev.DeviceVector(n)
            rc = dev.lu_solve(dev.DeviceCSR.from_scipy(A), dev.DeviceVector(data=b), x)
            assert rc == 0, "status %d" % rc
            assert dev.prof_get(8)[1] == c0 + (1 if kl >= 8 and n >= 64 else 0), "not on the expected path"      # (tiny systems: the LU)
            xs = x.get_local()
            res = np.linalg.norm(A @ xs - b) / np.linalg.norm(b)
            res_ref = np.linalg.norm(A @ ref - b) / np.linalg.norm(b)
            assert np.all(np.isfinite(xs)) and res <= max(1e-12, 50 * res_ref), "residual %.2e (LAPACK %.2e)" % (res, res_ref)
            assert np.max(np.abs(xs - ref)) <= 1e-9 * np.max(np.abs(ref)), "solution"
            if case % 5 == 0 and n > 8:
                C = A.copy().tolil()
                i = int(rng.integers(1, n - 1))
                if rng.random() < 0.5:
                    C[i, i] = -abs(C[i, i])                  # indefinite: a pivot that is not positive
                else:
                    C[i, i - 1] = C[i, i - 1] + 0.3 * abs(C[i, i])        # not symmetric
                C = C.tocsr(); C.sort_indices()
                bc = C @ nrhs_x
                c0 = dev.prof_get(8)[1]
                xc = dev.DeviceVector(n)
                rc = dev.lu_solve(dev.DeviceCSR.from_scipy(C), dev.DeviceVector(data=bc), xc)
                assert rc == 0, "handed on: status %d" % rc
                assert dev.prof_get(8)[1] == c0, "a matrix that is not SPD stayed on the Cholesky path"
                assert np.linalg.norm(C @ xc.get_local() - bc) <= 1e-8 * np.linalg.norm(bc), "handed on: residual"
        except AssertionError as e:
            bad += 1
            desc["error"] = str(e)
            print(json.dumps(desc), flush=True)
    print(json.dumps({"cases": args.cases, "failed": bad, "seed": args.seed}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
