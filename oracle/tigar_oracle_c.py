"""
TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never imported by the product path (tigar_amd/).

ctypes front end of oracle/tigar_oracle_c.c (C + OpenMP restatement of generateM, M^T A M with
MatZeroRowsColumns, M^T b and Jacobi-CG: the CSR algorithms PETSc AIJ runs on the CPU [ext]).  The 1-D
span/basis tables come from the scalar numpy restatement (``tigar_oracle._eval_1d_table``), which is
pinned against the reference's golden vectors; the C code is pinned against the numpy oracle in
tests/test_oracle_c.py.  a-8...a-12: parity unpinned against PETSc itself (absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

from . import tigar_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_i64p, _i32p, _f64p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "_build", "libtigar_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libtigar_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.tgo_num_threads.restype = C.c_int
        L.tgo_generate_M.restype = C.c_int64
        L.tgo_ptap.restype = C.c_int64
        L.tgo_ptap_blocked.restype = C.c_int64
        L.tgo_cg_jacobi.restype = C.c_int
        _LIB = L
    return _LIB


def num_threads():
    return int(lib().tgo_num_threads())


def usable_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (a container
    that sees 256 hardware threads but is throttled to a few makes 256 OpenMP threads crawl at every
    barrier)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def set_threads(n):
    lib().tgo_set_threads(int(n))
    return num_threads()


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def generate_M_tensor(bspline, degree=None, eps=O.DEFAULT_BASIS_FUNC_IGNORE_EPS):
    """generateM on the canonical tensor node grid (scipy CSR)."""
    L = lib()
    deg = bspline.getDegree() if degree is None else degree
    d = bspline.nvar
    axes = [O.fe_nodes_1d(s, deg, False) for s in bspline.splines]
    tabs = [O._eval_1d_table(bspline.splines[k], axes[k]) for k in range(d)]
    idx = [np.ascontiguousarray(t[0], dtype=np.int32) for t in tabs]
    val = [np.ascontiguousarray(t[1], dtype=np.float64) for t in tabs]
    n = np.array([len(a) for a in axes], dtype=np.int64)
    p1 = np.array([s.p + 1 for s in bspline.splines], dtype=np.int32)
    ncp = np.array([s.getNcp() for s in bspline.splines], dtype=np.int64)
    nrows = int(np.prod(n))
    idx_arr = (_i32p * d)(*[_p(a, _i32p) for a in idx])
    val_arr = (_f64p * d)(*[_p(a, _f64p) for a in val])
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    nnz = L.tgo_generate_M(d, _p(n, _i64p), _p(p1, _i32p), idx_arr, val_arr, _p(ncp, _i64p), C.c_double(eps),
                           _p(rowptr, _i64p), None, None)
    col = np.empty(max(nnz, 1), dtype=np.int32)
    v = np.empty(max(nnz, 1), dtype=np.float64)
    L.tgo_generate_M(d, _p(n, _i64p), _p(p1, _i32p), idx_arr, val_arr, _p(ncp, _i64p), C.c_double(eps),
                     _p(rowptr, _i64p), _p(col, _i32p), _p(v, _f64p))
    return sp.csr_matrix((v[:nnz], col[:nnz], rowptr), shape=(nrows, int(np.prod(ncp))))


def _csr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int64), np.ascontiguousarray(A.indices, dtype=np.int32),
            np.ascontiguousarray(A.data, dtype=np.float64))


def extract_matrix(M, A, zero_dofs, diag=1.0, max_am_entries=None):
    """M^T A M + MatZeroRowsColumns(zero_dofs, diag) (scipy CSR).  ``max_am_entries``: bound on the entries of the
    intermediate A*M held at a time (rows of K in blocks, ``tgo_ptap_blocked``); None: one block."""
    L = lib()
    if max_am_entries is not None:
        return _extract_matrix_blocked(M, A, zero_dofs, diag, int(max_am_entries))
    mrp, mci, mcv = _csr(M)
    arp, aci, acv = _csr(A)
    nfe, ncp = M.shape
    zd = np.ascontiguousarray(zero_dofs, dtype=np.int32)
    krp = np.zeros(ncp + 1, dtype=np.int64)
    args = (C.c_int64(nfe), C.c_int64(ncp), _p(mrp, _i64p), _p(mci, _i32p), _p(mcv, _f64p), _p(arp, _i64p), _p(aci, _i32p),
            _p(acv, _f64p), _p(zd, _i32p), C.c_int64(zd.size), C.c_double(diag))
    nnz = L.tgo_ptap(*args, _p(krp, _i64p), None, None)
    kc = np.empty(max(nnz, 1), dtype=np.int32)
    kv = np.empty(max(nnz, 1), dtype=np.float64)
    rc = L.tgo_ptap(*args, _p(krp, _i64p), _p(kc, _i32p), _p(kv, _f64p))
    assert rc == 0
    return sp.csr_matrix((kv[:nnz], kc[:nnz], krp), shape=(ncp, ncp))


def _extract_matrix_blocked(M, A, zero_dofs, diag, max_am_entries):
    L = lib()
    mrp, mci, mcv = _csr(M)
    arp, aci, acv = _csr(A)
    nfe, ncp = M.shape
    zd = np.ascontiguousarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
    krp = np.zeros(ncp + 1, dtype=np.int64)
    args = (C.c_int64(nfe), C.c_int64(ncp), _p(mrp, _i64p), _p(mci, _i32p), _p(mcv, _f64p), _p(arp, _i64p), _p(aci, _i32p),
            _p(acv, _f64p), _p(zd, _i32p), C.c_int64(zd.size), C.c_double(diag), C.c_int64(max_am_entries))
    nnz = L.tgo_ptap_blocked(*args, _p(krp, _i64p), None, None)
    kc = np.empty(max(nnz, 1), dtype=np.int32)
    kv = np.empty(max(nnz, 1), dtype=np.float64)
    rc = L.tgo_ptap_blocked(*args, _p(krp, _i64p), _p(kc, _i32p), _p(kv, _f64p))
    assert rc == 0
    return sp.csr_matrix((kv[:nnz], kc[:nnz], krp), shape=(ncp, ncp))


def ptap_sum_factorised(M1, A, zero_dofs, diag=1.0, max_am_entries=None):
    """K = P_z^T (P_y^T (P_x^T A P_x) P_y) P_z with P_k = I (x) M_k (x) I, every stage by the same C Gustavson products
    as ``extract_matrix`` (the GPU path's algorithm restated on the CPU; M1: the 1-D extraction matrices, direction 0
    first).  The boundary conditions go with the last stage."""
    d = len(M1)
    cur = sp.csr_matrix(A)
    for k in range(d):
        dims = [M1[j].shape[1] if j < k else M1[j].shape[0] for j in range(d)]
        facs = [sp.csr_matrix(M1[j]) if j == k else sp.identity(dims[j], format="csr") for j in range(d)]
        Pk = O.kron_dir0_fastest(facs).tocsr()
        last = k == d - 1
        cur = extract_matrix(Pk, cur, zero_dofs if last else [], diag, max_am_entries)
    return cur


def extract_vector(M, b, zero_dofs):
    L = lib()
    mrp, mci, mcv = _csr(M)
    b = np.ascontiguousarray(b, dtype=np.float64)
    zd = np.ascontiguousarray(zero_dofs, dtype=np.int32)
    y = np.empty(M.shape[1])
    L.tgo_spmv_t(C.c_int64(M.shape[0]), C.c_int64(M.shape[1]), _p(mrp, _i64p), _p(mci, _i32p), _p(mcv, _f64p), _p(b, _f64p),
                 _p(zd, _i32p), C.c_int64(zd.size), _p(y, _f64p))
    return y


def spmv(A, x):
    L = lib()
    rp, ci, cv = _csr(A)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[0])
    L.tgo_spmv(C.c_int64(A.shape[0]), _p(rp, _i64p), _p(ci, _i32p), _p(cv, _f64p), _p(x, _f64p), _p(y, _f64p))
    return y


def cg_jacobi(K, b, rtol=1e-6, atol=1e-15, maxit=10000):
    """(x, iterations, preconditioned residual norm); iterations < 0: not converged."""
    L = lib()
    rp, ci, cv = _csr(K)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(K.shape[0])
    res = C.c_double()
    its = L.tgo_cg_jacobi(C.c_int64(K.shape[0]), _p(rp, _i64p), _p(ci, _i32p), _p(cv, _f64p), _p(b, _f64p), C.c_double(rtol),
                          C.c_double(atol), C.c_int(maxit), _p(x, _f64p), C.byref(res))
    return x, its, res.value
