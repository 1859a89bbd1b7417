"""Randomised run of the sparse kernels behind the C-ABI against scipy (developer tool; `tests/test_gpu_fuzz.py` runs a seeded
set): matrices of arbitrary shape and density -- empty rows and columns, a few very long rows, one-row / one-column / empty
matrices -- through

* SpMV (CSR kernels and the sliced copy), M^T b, transpose, add, column selection / permutation, row gather, blocks;
* the general PtAP K = M^T A M (tIGAr/common.py:1194-1195 applies it to ANY A and M): structural pattern and values, the
  second product on the plan bit-identical, MatZeroRowsColumns fused;
* the Krylov solvers on diagonally dominant systems.

    python tests/fuzz/fuzz_kernels.py [--seed S] [--cases N] [-v]

The kernel family follows the environment (TIGAR_PTAP_WAVE, TIGAR_PTAP_ACCUM, TIGAR_KSP_PERSISTENT, TIGAR_POOL_POISON)."""
import argparse
import json
import os
import sys
import traceback

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def rand_csr(rng, n, m, kind):
    """a random n x m CSR matrix; `kind` picks the row-length profile"""
    if n == 0 or m == 0:
        return sp.csr_matrix((n, m))
    if kind == "uniform":
        dens = float(rng.choice([0.002, 0.02, 0.1, 0.5]))
        lens = rng.binomial(m, min(1.0, dens), size=n)
    elif kind == "ragged":                      # most rows empty, a few long ones
        lens = np.where(rng.random(n) < 0.7, 0, rng.integers(1, max(2, min(m, 400)), size=n))
    elif kind == "banded":
        lens = np.minimum(m, rng.integers(1, 12, size=n))
    else:                                       # "long": one or two rows that fill the matrix
        lens = rng.integers(0, 4, size=n)
        lens[rng.integers(0, n, size=2)] = m
    lens = np.minimum(lens, m)
    rows, cols = [], []
    for i, l in enumerate(lens):
        if l:
            if kind == "banded":
                c0 = int(rng.integers(0, max(1, m - l + 1)))
                c = np.arange(c0, c0 + l)
            else:
                c = rng.choice(m, size=int(l), replace=False)
            rows.append(np.full(int(l), i))
            cols.append(c)
    if not rows:
        return sp.csr_matrix((n, m))
    r, c = np.concatenate(rows), np.concatenate(cols)
    v = rng.standard_normal(r.size) * (10.0 ** rng.integers(-2, 3))
    A = sp.csr_matrix((v, (r, c)), shape=(n, m))
    A.sort_indices()
    return A


def structural(M, A):
    def ones(X):
        X = sp.csr_matrix(X, copy=True)
        X.data = np.ones(X.nnz)
        return X
    S = (ones(M).T @ ones(A) @ ones(M)).tocsr()
    S.sort_indices()
    return S


def same(Kd, Ko, tol, what):
    K = Kd.to_scipy().tocsr()
    K.sort_indices()
    Ko = sp.csr_matrix(Ko)
    Ko.sort_indices()
    assert K.shape == Ko.shape, "%s: shape %s != %s" % (what, K.shape, Ko.shape)
    scale = max(1e-300, abs(Ko).max() if Ko.nnz else 0.0)
    d = abs(K - Ko).max() if (K.nnz or Ko.nnz) else 0.0
    assert d <= tol * scale, "%s: values differ by %g of %g" % (what, d, scale)
    return K


def run_case(rng, verbose):
    from tigar_amd import device as dev
    n = int(rng.choice([0, 1, 2, 7, 64, 65, 300, 1500, 5000]))
    m = int(rng.choice([1, 3, 33, 128, 700, 2500]))
    kindA = str(rng.choice(["uniform", "ragged", "banded", "long"]))
    kindM = str(rng.choice(["uniform", "ragged", "banded", "long"]))
    A = rand_csr(rng, n, n, kindA)
    M = rand_csr(rng, n, m, kindM)
    # keep the product within what a test should take: cap the flops of the triple product
    work = float((abs(M).T @ (abs(A) @ abs(M).sum(axis=1))).sum()) if n and m and A.nnz and M.nnz else 0.0
    if verbose:
        print("  n %d m %d A %s nnz %d  M %s nnz %d" % (n, m, kindA, A.nnz, kindM, M.nnz), flush=True)
    Ad, Md = dev.DeviceCSR.from_scipy(A), dev.DeviceCSR.from_scipy(M)
    # ---- simple operations
    x = rng.standard_normal(m)
    y = Md.mult(dev.DeviceVector(data=x)).get_local() if n else np.zeros(0)
    assert np.allclose(y, M @ x, rtol=1e-12, atol=1e-12 * (1 + abs(M).max() if M.nnz else 1)), "M x"
    b = rng.standard_normal(n)
    z = Md.mult_transpose(dev.DeviceVector(data=b)).get_local()
    assert np.allclose(z, M.T @ b, rtol=1e-11, atol=1e-11 * (1 + (abs(M).max() if M.nnz else 0.0) * max(1.0, np.abs(b).max() if n else 1.0) * max(1, n))), "M^T b"
    MT = Md.transpose()
    same(MT, M.T, 0.0, "transpose")
    if n:
        A2 = rand_csr(rng, n, n, "uniform")
        same(Ad.add(dev.DeviceCSR.from_scipy(A2)), A + A2, 1e-15, "add")
        keep = rng.random(n) < 0.5
        Sel = A @ sp.diags(keep.astype(float))
        Sel.eliminate_zeros()
        K = same(Ad.select_columns(keep), Sel, 0.0, "select_columns")
        assert K.nnz == Sel.nnz, "select_columns keeps entries it should drop"
        perm = rng.permutation(n)
        P = sp.csr_matrix((np.ones(n), (np.arange(n), perm)), shape=(n, n))
        same(Ad.permute_columns(perm), A @ P, 0.0, "permute_columns")
        rows = rng.integers(0, n, size=min(n, 17))
        same(Ad.gather_rows(rows), A[rows], 0.0, "gather_rows")
        r0, r1 = sorted(rng.integers(0, n + 1, size=2).tolist())
        c0, c1 = sorted(rng.integers(0, n + 1, size=2).tolist())
        same(Ad.block(r0, r1, c0, c1), A[r0:r1, c0:c1], 0.0, "block")
        xs = rng.standard_normal(n)
        ya = Ad.mult(dev.DeviceVector(data=xs)).get_local()
        assert np.allclose(ya, A @ xs, rtol=1e-12, atol=1e-12 * (1 + (abs(A).max() if A.nnz else 0) * n)), "A x"
    # ---- the general PtAP
    if work <= 3e8 and m > 0:
        Ko = (M.T @ A @ M).tocsr()
        S = structural(M, A)
        zd = sorted(set(rng.integers(0, m, size=int(rng.integers(0, 4))).tolist()))
        diag = float(rng.choice([1.0, 2.5]))
        plan = dev.ptap_symbolic(Ad, Md, MT)
        Kd = dev.ptap_numeric(plan, Ad, Md, MT, zd if zd else None, diag)
        # MatZeroRowsColumns keeps the pattern: `diag` goes where the structural product has a diagonal entry (PETSc
        # refuses a matrix without one [ext]; the K of a spline space always has it), as oracle.zero_rows_columns does
        Kref = Ko.tolil()
        for i in zd:
            Kref[i, :] = 0.0
            Kref[:, i] = 0.0
            if S[i, i] != 0:
                Kref[i, i] = diag
        K = same(Kd, Kref.tocsr(), 1e-12, "PtAP")
        pat = S
        assert np.array_equal(K.indptr, pat.indptr) and np.array_equal(K.indices, pat.indices), \
            "pattern of PtAP: %d stored, %d structural" % (K.nnz, pat.nnz)
        K2 = dev.ptap_numeric(plan, Ad, Md, MT, zd if zd else None, diag).to_scipy()
        K2.sort_indices()
        if not np.array_equal(K2.data.view(np.int64), K.data.view(np.int64)):
            # the fused kernel adds integers on a per-row grid unless the operand rows of a K row differ in scale by more
            # than 2^16 (such rows accumulate in floating point, accurate per entry, last bits free);
            # random values produce such rows now and then.  With TIGAR_PTAP_ACCUM=int every row is on the grid.
            assert os.environ.get("TIGAR_PTAP_ACCUM") not in ("int",) and os.environ.get("TIGAR_PTAP_WAVE") != "1", \
                "PtAP not bit-reproducible"
            assert abs(K2 - K).max() <= 1e-13 * max(1e-300, abs(K).max()), "PtAP differs between two runs"
    # ---- Krylov solvers on a diagonally dominant system
    if 1 <= n <= 5000:
        B = rand_csr(rng, n, n, str(rng.choice(["uniform", "banded"])))
        symmetric = rng.random() < 0.5
        if symmetric:
            B = (B + B.T).tocsr()
        B = (B + sp.diags(np.asarray(abs(B).sum(axis=1)).ravel() * 1.5 + 1.0)).tocsr()
        B.sort_indices()
        rhs = rng.standard_normal(n)
        xo = spla.spsolve(B.tocsc(), rhs) if n > 1 else rhs / B[0, 0]
        Bd = dev.DeviceCSR.from_scipy(B)
        for method, pc in ([("cg", "jacobi"), ("cg", "chebyshev"), ("cg", "none")] if symmetric else []) + \
                [("gmres", "jacobi"), ("bicgstab", "jacobi"), ("gmres", "none")]:
            xd = dev.DeviceVector(n)
            its, res, status = dev.krylov_solve(Bd, dev.DeviceVector(data=rhs), xd, method, pc, 1e-12, 1e-300, 5000, 30)
            e = np.max(np.abs(xd.get_local() - xo)) / max(1e-300, np.max(np.abs(xo)))
            assert status == 0 and e <= 1e-8, "%s/%s: status %d, error %g after %d iterations" % (method, pc, status, e, its)
        xs = rng.standard_normal(n)
        Bd.spmv_sell(True)
        ys = Bd.mult(dev.DeviceVector(data=xs)).get_local()
        assert np.allclose(ys, B @ xs, rtol=1e-12, atol=1e-12 * abs(B).max() * n), "sliced SpMV"


def main():
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    bad = 0
    for i in range(a.first + a.cases):
        rng = np.random.default_rng([a.seed, i])
        if i < a.first:
            continue
        if a.v:
            print("case %d" % i, flush=True)
        try:
            run_case(rng, a.v)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(json.dumps({"failed": i, "seed": a.seed, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}), flush=True)
            if a.v:
                traceback.print_exc()
    print(json.dumps({"cases": a.cases, "failed": bad, "seed": a.seed}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
