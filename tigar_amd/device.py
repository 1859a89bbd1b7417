"""
Thin object layer over the C-ABI: device-resident vectors / CSR matrices and the kernels of
the extraction path.  These are the objects the tIGAr-compatible API (``tigar_amd.common``)
hands around in place of dolfin's PETScMatrix / PETScVector.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import TigarHipError, check, handle, tg_dir_t, tg_kron_dir_t, tg_kron1d_t, tg_patch_t, c_f64p, c_i32p, c_i64p

TG_KSP_CG, TG_KSP_GMRES, TG_KSP_BICGSTAB = 0, 1, 2
TG_PC_NONE, TG_PC_JACOBI, TG_PC_CHEBYSHEV = 0, 1, 2
TG_KSP_NONZERO_GUESS = 1
TG_KSP_STAGNATION_GUARD = 2
TG_KSP_SYMMETRIC = 4


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t)


class DeviceVector(object):
    """fp64 vector in HBM (stands in for dolfin PETScVector on this path)."""

    def __init__(self, n=None, data=None, _handle=None, zero=True):
        """``zero=False``: contents undefined (an output some kernel overwrites entirely; no fill pass)"""
        L = _lib.lib()
        self._h = handle()
        if _handle is not None:
            self._h = _handle
            return
        if data is not None:
            data = _f64(data)
            n = data.shape[0]
        if zero and data is None:
            check(L.tg_vec_create(int(n), C.byref(self._h)), "tg_vec_create")
        else:
            check(L.tg_vec_create_uninit(int(n), C.byref(self._h)), "tg_vec_create_uninit")
        if data is not None:
            check(L.tg_vec_upload(self._h, _p(data, c_f64p), n), "tg_vec_upload")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_vec_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def size(self):
        n = C.c_int64()
        check(_lib.lib().tg_vec_size(self._h, C.byref(n)))
        return n.value

    __len__ = size

    # dolfin GenericVector-style accessors
    def get_local(self):
        out = np.empty(self.size(), dtype=np.float64)
        check(_lib.lib().tg_vec_download(self._h, _p(out, c_f64p), out.shape[0]), "tg_vec_download")
        return out

    def set_local(self, values):
        values = _f64(values)
        check(_lib.lib().tg_vec_upload(self._h, _p(values, c_f64p), values.shape[0]), "tg_vec_upload")

    to_numpy = get_local

    def copy(self):
        v = DeviceVector(self.size())
        check(_lib.lib().tg_vec_copy(v._h, self._h))
        return v

    def zero(self):
        check(_lib.lib().tg_vec_fill(self._h, 0.0))

    def fill(self, a):
        check(_lib.lib().tg_vec_fill(self._h, float(a)))

    def axpy(self, a, x):
        check(_lib.lib().tg_vec_axpy(self._h, float(a), x._h))

    def pointwise_mult(self, other, out=None):
        """out = self .* other (PETSc VecPointwiseMult)"""
        out = DeviceVector(self.size()) if out is None else out
        check(_lib.lib().tg_vec_pointwise_mult(out._h, self._h, other._h), "tg_vec_pointwise_mult")
        return out

    def inner(self, other):
        out = C.c_double()
        check(_lib.lib().tg_vec_dot(self._h, other._h, C.byref(out)))
        return out.value

    def norm(self, kind="l2"):
        """dolfin GenericVector.norm: "l1", "l2" or "linf"."""
        kinds = {"l1": 0, "l2": 1, "linf": 2}
        if kind not in kinds:
            raise ValueError("unsupported norm type %r (l1, l2, linf)" % (kind,))
        out = C.c_double()
        check(_lib.lib().tg_vec_norm(self._h, kinds[kind], C.byref(out)), "tg_vec_norm")
        return out.value

    def zero_entries(self, dofs, g0=0):
        """y[d - g0] = 0 for the global dofs d that fall into this (slab-local) vector"""
        dofs = _i32(dofs)
        if dofs.size:
            check(_lib.lib().tg_vec_zero_entries_offset(self._h, _p(dofs, c_i32p), dofs.size, int(g0)))


class DeviceCSR(object):
    """CSR row block in HBM: int64 rowptr, int32 (global) columns, fp64 values."""

    def __init__(self, _handle):
        self._h = _handle
        self._T = None          # cached explicit transpose

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_csr_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def from_scipy(A):
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        A.sort_indices()
        rowptr = _i64(A.indptr)
        col = _i32(A.indices)
        val = _f64(A.data)
        # (the kernels trust the structure: a column outside the matrix would be read as an address)
        if col.size and (int(col.min()) < 0 or int(col.max()) >= A.shape[1]):
            raise ValueError("DeviceCSR.from_scipy: column index outside [0, %d)" % A.shape[1])
        if rowptr.size != A.shape[0] + 1 or int(rowptr[0]) != 0 or int(rowptr[-1]) != col.size or np.any(np.diff(rowptr) < 0):
            raise ValueError("DeviceCSR.from_scipy: row pointer is not a non-decreasing array from 0 to nnz")
        h = handle()
        check(_lib.lib().tg_csr_from_host(A.shape[0], A.shape[1], _p(rowptr, c_i64p), _p(col, c_i32p),
                                          _p(val, c_f64p), C.byref(h)), "tg_csr_from_host")
        return DeviceCSR(h)

    @property
    def shape(self):
        r, c, z = C.c_int64(), C.c_int64(), C.c_int64()
        check(_lib.lib().tg_csr_dims(self._h, C.byref(r), C.byref(c), C.byref(z)))
        return (r.value, c.value)

    @property
    def nnz(self):
        r, c, z = C.c_int64(), C.c_int64(), C.c_int64()
        check(_lib.lib().tg_csr_dims(self._h, C.byref(r), C.byref(c), C.byref(z)))
        return z.value

    def to_scipy(self):
        import scipy.sparse as sp
        (nr, nc), nnz = self.shape, self.nnz
        rowptr = np.empty(nr + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.float64)
        check(_lib.lib().tg_csr_download(self._h, _p(rowptr, c_i64p), _p(col, c_i32p), _p(val, c_f64p)),
              "tg_csr_download")
        return sp.csr_matrix((val, col, rowptr), shape=(nr, nc))

    def rows_to_scipy(self, r0, r1):
        """Rows [r0, r1) as a scipy CSR block (r1-r0 x ncols) -- for matrices too large to download whole."""
        import scipy.sparse as sp
        n = int(r1) - int(r0)
        rowptr = np.empty(n + 1, dtype=np.int64)
        check(_lib.lib().tg_csr_download_rows(self._h, int(r0), int(r1), _p(rowptr, c_i64p), None, None, 0),
              "tg_csr_download_rows")
        cnt = int(rowptr[-1])
        col = np.empty(max(cnt, 1), dtype=np.int32)
        val = np.empty(max(cnt, 1), dtype=np.float64)
        check(_lib.lib().tg_csr_download_rows(self._h, int(r0), int(r1), _p(rowptr, c_i64p), _p(col, c_i32p),
                                              _p(val, c_f64p), col.shape[0]), "tg_csr_download_rows")
        return sp.csr_matrix((val[:cnt], col[:cnt], rowptr), shape=(n, self.shape[1]))

    def combine(self, a, other, b, colscale=None):
        """a*self + b*other*diag(colscale) for two matrices on one pattern (checked on the device)"""
        h = handle()
        check(_lib.lib().tg_csr_combine(float(a), self._h, float(b), other._h,
                                        colscale._h if colscale is not None else None, C.byref(h)), "tg_csr_combine")
        return DeviceCSR(h)

    def rowptr_at(self, r):
        v = C.c_int64()
        check(_lib.lib().tg_csr_rowptr_at(self._h, int(r), C.byref(v)), "tg_csr_rowptr_at")
        return v.value

    def is_loose(self):
        v = C.c_int()
        check(_lib.lib().tg_csr_is_loose(self._h, C.byref(v)), "tg_csr_is_loose")
        return bool(v.value)

    def compact(self):
        """canonical CSR copy of a loose-row stage result (``tg_csr_compact``)"""
        h = handle()
        check(_lib.lib().tg_csr_compact(self._h, C.byref(h)), "tg_csr_compact")
        return DeviceCSR(h)

    def transpose(self):
        if self._T is None:
            h = handle()
            check(_lib.lib().tg_csr_transpose(self._h, C.byref(h)), "tg_csr_transpose")
            self._T = DeviceCSR(h)
        return self._T

    def add(self, other):
        """self + other on the union of the two patterns (tg_csr_add)"""
        h = handle()
        check(_lib.lib().tg_csr_add(self._h, other._h, C.byref(h)), "tg_csr_add")
        return DeviceCSR(h)

    def block(self, r0, r1, c0, c1):
        """rows [r0, r1) x columns [c0, c1) as a matrix of its own, columns renumbered from 0 (tg_csr_block)"""
        h = handle()
        check(_lib.lib().tg_csr_block(self._h, int(r0), int(r1), int(c0), int(c1), C.byref(h)), "tg_csr_block")
        return DeviceCSR(h)

    def select_columns(self, keep):
        """copy of the same shape without the entries of the columns whose ``keep`` flag is False (tg_csr_select_columns)"""
        k = np.ascontiguousarray(np.asarray(keep) != 0, dtype=np.uint8)
        if k.size != self.shape[1]:
            raise ValueError("select_columns: %d flags for %d columns" % (k.size, self.shape[1]))
        h = handle()
        check(_lib.lib().tg_csr_select_columns(self._h, k.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(h)),
              "tg_csr_select_columns")
        return DeviceCSR(h)

    def gather_rows(self, rows):
        """new matrix whose row r is row ``rows[r]`` of this one (tg_csr_gather_rows)"""
        m = np.ascontiguousarray(rows, dtype=np.int64)
        h = handle()
        check(_lib.lib().tg_csr_gather_rows(self._h, m.ctypes.data_as(c_i64p), m.size, C.byref(h)), "tg_csr_gather_rows")
        return DeviceCSR(h)

    def permute_columns(self, new_of_old):
        """copy with column c renamed ``new_of_old[c]`` and rows re-sorted (MatPermute with identity rows)"""
        m = np.ascontiguousarray(new_of_old, dtype=np.int32)
        if m.size != self.shape[1]:
            raise ValueError("permute_columns: %d entries for %d columns" % (m.size, self.shape[1]))
        h = handle()
        check(_lib.lib().tg_csr_permute_columns(self._h, m.ctypes.data_as(c_i32p), C.byref(h)), "tg_csr_permute_columns")
        return DeviceCSR(h)

    def majority_owner(self, fe_owner, world):
        """for every row (an IGA dof of a transposed extraction pattern) the rank owning most of its columns
        (FE rows, owner given per column); lowest rank on ties (tg_partition_mode)"""
        own = np.ascontiguousarray(fe_owner, dtype=np.int32)
        if own.size != self.shape[1]:
            raise ValueError("majority_owner: %d owners for %d columns" % (own.size, self.shape[1]))
        out = np.zeros(self.shape[0], dtype=np.int32)
        check(_lib.lib().tg_partition_mode(self._h, own.ctypes.data_as(c_i32p), int(world), out.ctypes.data_as(c_i32p)),
              "tg_partition_mode")
        return out

    def mult(self, x, y=None):
        """y = A x"""
        if y is None:
            y = DeviceVector(self.shape[0])
        check(_lib.lib().tg_spmv(self._h, x._h, y._h), "tg_spmv")
        return y

    def spmv_sell(self, enable=True):
        """Keep (or drop) the sliced, pattern-compressed copy of the values for repeated products
        (tg_spmv_sell; a snapshot -- request it again after changing values).  Returns (number of slice
        classes, doubles stored); (0, 0) if the matrix has no such structure or enable is False."""
        n, padded = C.c_int(0), C.c_int64(0)
        check(_lib.lib().tg_spmv_sell(self._h, 1 if enable else 0, C.byref(n), C.byref(padded)), "tg_spmv_sell")
        return n.value, padded.value

    def mult_symgrid(self, x=None, y=None, row0=0):
        """The half-storage product of the CG solve (tg_spmv_symgrid): plans the copy, checks it against the CSR product
        and, with x given, returns (y, info); ``row0``: this matrix holds the rows [row0, row0 + nrows) of a square one (a z
        slab, all columns; x covers all columns).  info = None when the matrix is not a symmetric 3-D box stencil (y is then
        None as well), else {"value_bytes": bytes of K one product reads, "staging_bytes": ...}."""
        ok, vb, sb = C.c_int(0), C.c_int64(0), C.c_int64(0)
        if x is not None and y is None:
            y = DeviceVector(self.shape[0])
        check(_lib.lib().tg_spmv_symgrid(self._h, int(row0), x._h if x is not None else None, y._h if x is not None else None,
                                         C.byref(ok), C.byref(vb), C.byref(sb)), "tg_spmv_symgrid")
        if not ok.value:
            return None, None
        return y, {"value_bytes": vb.value, "staging_bytes": sb.value}

    def mult_offset(self, x, x_col0, y=None):
        """y = A x where x holds only the columns [x_col0, x_col0+len(x)) (slab pieces)"""
        if y is None:
            y = DeviceVector(self.shape[0])
        check(_lib.lib().tg_spmv_offset(self._h, x._h, int(x_col0), y._h), "tg_spmv_offset")
        return y

    def __mul__(self, x):
        if isinstance(x, DeviceVector):
            return self.mult(x)
        return NotImplemented

    def mult_transpose(self, b, y=None):
        """y = A^T b through the explicit transpose (scatter-free)."""
        T = self.transpose()
        if y is None:
            y = DeviceVector(T.shape[0])
        check(_lib.lib().tg_spmv_t(T._h, b._h, y._h), "tg_spmv_t")
        return y

    def spmm_host(self, X):
        X = np.asfortranarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1, order="F")
        k = X.shape[1]
        Y = np.empty((self.shape[0], k), dtype=np.float64, order="F")
        check(_lib.lib().tg_spmm_host(self._h, _p(X, c_f64p), k, _p(Y, c_f64p)), "tg_spmm_host")
        return Y

    def zero_rows_cols(self, dofs, diag=1.0, row0=0):
        dofs = _i32(dofs)
        if dofs.size:
            check(_lib.lib().tg_zero_rows_cols(self._h, int(row0), _p(dofs, c_i32p), dofs.size, float(diag)),
                  "tg_zero_rows_cols")


# ------------------------------------------------------------------------------- extraction
class _DirPack(object):
    """Keeps the numpy buffers of a tg_dir_t array alive."""

    def __init__(self, splines1d, nodes1d):
        self.keep = []
        self.arr = (tg_dir_t * len(splines1d))()
        for k, (s, x) in enumerate(zip(splines1d, nodes1d)):
            ghost = _f64(s.ghostKnots)
            x = _f64(x) if x is not None else np.zeros(0)
            self.keep += [ghost, x]
            d = self.arr[k]
            d.p = int(s.p)
            d.nknots = int(len(s.knots))
            d.ghost = _p(ghost, c_f64p)
            d.mult_first = int(s.multiplicities[0])
            d.mult_last = int(s.multiplicities[-1])
            d.ncp = int(s.ncp)
            d.nnodes = int(x.shape[0])
            d.nodes = _p(x, c_f64p)


def extract_csr_tensor(splines1d, nodes1d, col_offset, ncols, eps, row0=None, row1=None):
    """generateM for a BSpline on the implicit tensor node grid (kernel path)."""
    pack = _DirPack(splines1d, nodes1d)
    total = 1
    for x in nodes1d:
        total *= len(x)
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    h = handle()
    check(_lib.lib().tg_extract_csr_tensor(len(splines1d), pack.arr, int(col_offset), int(ncols), float(eps),
                                           row0, row1, C.byref(h)), "tg_extract_csr_tensor")
    return DeviceCSR(h)


def extract_csr_tensor_t(splines1d, nodes1d, fe_row_offset, fe_rows_total, eps, dof0=None, dof1=None):
    """M^T of ``extract_csr_tensor`` written directly (rows = spline dofs [dof0,dof1))."""
    pack = _DirPack(splines1d, nodes1d)
    ncp = 1
    for s in splines1d:
        ncp *= int(s.ncp)
    dof0 = 0 if dof0 is None else int(dof0)
    dof1 = ncp if dof1 is None else int(dof1)
    h = handle()
    check(_lib.lib().tg_extract_csr_tensor_t(len(splines1d), pack.arr, int(fe_row_offset), int(fe_rows_total),
                                             float(eps), dof0, dof1, C.byref(h)), "tg_extract_csr_tensor_t")
    return DeviceCSR(h)


def extract_apply_tensor(splines1d, nodes1d, col_offset, eps, x, x_col0=0, row0=None, row1=None, y=None):
    """y = M x without forming M (same arithmetic as extract_csr_tensor followed by an SpMV)."""
    pack = _DirPack(splines1d, nodes1d)
    total = 1
    for a in nodes1d:
        total *= len(a)
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    if y is None:
        y = DeviceVector(row1 - row0)
    check(_lib.lib().tg_extract_apply_tensor(len(splines1d), pack.arr, int(col_offset), float(eps), row0, row1,
                                             x._h, int(x_col0), y._h), "tg_extract_apply_tensor")
    return y


def extract_csr_points(splines1d, x, col_offset, ncols, eps):
    x = _f64(x)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    pack = _DirPack(splines1d, [None] * len(splines1d))
    h = handle()
    check(_lib.lib().tg_extract_csr_points(len(splines1d), pack.arr, int(col_offset), int(ncols), float(eps),
                                           _p(x, c_f64p), x.shape[0], C.byref(h)), "tg_extract_csr_points")
    return DeviceCSR(h)


def extract_csr_bezier(bern, eoff, nodes, coef, col_offset, ncols, eps):
    """Extraction rows of a spline given by element-wise Bezier extraction operators (``tg_extract_csr_bezier``):
    ``bern[e, n, b]`` Bernstein values at FE node n of element e, ``eoff`` offsets of the elements in ``nodes`` (global
    function indices, ascending per element) and ``coef`` (their extraction rows)."""
    bern, coef = _f64(bern), _f64(coef)
    nel, nloc, nbern = bern.shape
    eoff, nodes = _i64(eoff), _i32(nodes)
    h = handle()
    check(_lib.lib().tg_extract_csr_bezier(int(nel), int(nloc), int(nbern), _p(bern, c_f64p), _p(eoff, c_i64p),
                                           _p(nodes, c_i32p), _p(coef, c_f64p), int(col_offset), int(ncols), float(eps),
                                           C.byref(h)), "tg_extract_csr_bezier")
    return DeviceCSR(h)


def csr_from_blocks(blocks):
    """nf x nf blocks (list of rows of DeviceCSR; block (i, j) has the rows of field i and the columns of field j) -> the matrix
    with field-major rows and columns"""
    nf = len(blocks)
    flat = [b for row in blocks for b in row]
    if any(len(row) != nf for row in blocks):
        raise ValueError("csr_from_blocks: a square arrangement of blocks is expected")
    arr = (handle * len(flat))(*[b._h for b in flat])
    h = handle()
    check(_lib.lib().tg_csr_from_blocks(nf, arr, C.byref(h)), "tg_csr_from_blocks")
    return DeviceCSR(h)


def csr_vstack(blocks):
    arr = (handle * len(blocks))(*[b._h for b in blocks])
    h = handle()
    check(_lib.lib().tg_csr_vstack(len(blocks), arr, C.byref(h)), "tg_csr_vstack")
    return DeviceCSR(h)


def csr_vstack_view(blocks):
    """Stacked loose-row VIEW of the blocks (no entry copies, ``tg_csr_vstack_view``); the returned
    object keeps the blocks alive."""
    arr = (handle * len(blocks))(*[b._h for b in blocks])
    h = handle()
    check(_lib.lib().tg_csr_vstack_view(len(blocks), arr, C.byref(h)), "tg_csr_vstack_view")
    out = DeviceCSR(h)
    out._keep = list(blocks)
    return out


class CSRBuilder(object):
    """Incremental vstack of row blocks into one allocation (K assembled slab by slab)."""

    def __init__(self, nrows_total, ncols, nnz_capacity):
        self._h = handle()
        check(_lib.lib().tg_csr_builder_create(int(nrows_total), int(ncols), int(nnz_capacity), C.byref(self._h)),
              "tg_csr_builder_create")

    def append(self, block):
        check(_lib.lib().tg_csr_builder_append(self._h, block._h), "tg_csr_builder_append")

    def finish(self):
        h = handle()
        check(_lib.lib().tg_csr_builder_finish(self._h, C.byref(h)), "tg_csr_builder_finish")
        self._h = None
        return DeviceCSR(h)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_csr_builder_destroy(self._h)
                self._h = None
        except Exception:
            pass


def csr_from_triplets(nrows, ncols, rows, cols, vals, eps):
    rows, cols, vals = _i64(rows), _i32(cols), _f64(vals)
    h = handle()
    check(_lib.lib().tg_csr_from_triplets(int(nrows), int(ncols), rows.size, _p(rows, c_i64p), _p(cols, c_i32p),
                                          _p(vals, c_f64p), float(eps), C.byref(h)), "tg_csr_from_triplets")
    return DeviceCSR(h)


def eval_basis_1d(spline1, u):
    """Device twin of BSpline1.getKnotSpan / getNodes / basisFuncs for an array of u."""
    u = _f64(u)
    pack = _DirPack([spline1], [None])
    n = u.shape[0]
    span = np.empty(n, dtype=np.int32)
    idx = np.empty((n, spline1.p + 1), dtype=np.int32)
    val = np.empty((n, spline1.p + 1), dtype=np.float64)
    check(_lib.lib().tg_eval_basis_1d(pack.arr, _p(u, c_f64p), n, _p(span, c_i32p), _p(idx, c_i32p),
                                      _p(val, c_f64p)), "tg_eval_basis_1d")
    return span, idx, val


# ------------------------------------------------------------------------------- PtAP
class PtAPPlan(object):
    def __init__(self, h):
        self._h = h

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_ptap_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _capacity_error(e):
    """a row of the product (or of A M) exceeds the per-row LDS tables of the general kernels"""
    msg = str(e)
    return "too dense for the LDS tables" in msg or "LDS hash slots" in msg or "could not size tables" in msg


class SplitPtAPPlan(object):
    """M^T A M as a sum of partial products over residue classes of columns:

        K = sum_t sum_r  M^T A_t M_r      A_t = the columns c = t (mod nt) of A,  M_r = the columns c = r (mod nr) of M

    (``tg_csr_select_columns``; the classes are disjoint, so the sums are unions).  A row of M^T A_t holds about 1/nt of the
    keys of a row of M^T A, a row of a partial product about 1/nr of a row of K: rows beyond the per-row LDS tables of the
    general kernels -- 3-D patches of degree >= 5, where a row of M^T A reaches (3p+1)^3 > 4096 FE nodes -- come back inside
    them.  The reference's MatPtAP has no degree limit (tIGAr/common.py:1194-1195); this is the route such products take
    here: nt x nr passes over the operands instead of an error.  The classes are kept, the selections of A are redone at
    every numeric product (its values change between calls)."""

    LADDER = ((4, 1), (1, 4), (16, 1), (4, 4), (64, 1), (16, 4), (64, 4), (256, 4), (64, 16), (256, 16), (1024, 16))

    def __init__(self, A, M, MT, a_row0, m_row0, mt_row0):
        self.rows = (int(a_row0), int(m_row0), int(mt_row0))
        self._first_of = A
        ca, cm = np.arange(A.shape[1]), np.arange(M.shape[1])
        last = None
        for nt, nr in self.LADDER:
            try:
                parts = []
                Ms = [(M.select_columns((cm % nr) == r) if nr > 1 else M) for r in range(nr)]
                for t_ in range(nt):
                    mask = (ca % nt) == t_
                    At = A.select_columns(mask) if nt > 1 else A
                    if At.nnz == 0:
                        continue
                    for Mr in Ms:
                        if Mr.nnz:
                            parts.append((mask if nt > 1 else None, Mr, _ptap_symbolic_plain(At, Mr, MT, *self.rows)))
                    del At
                # (the symbolic pass looks at a sample of rows: the numeric product of every part is tried once here, so
                #  that a plan that is handed out works)
                self.parts, self.split = parts, (nt, nr)
                self._first = self._numeric(A, MT)
                return
            except TigarHipError as e:
                if not _capacity_error(e):
                    raise
                last = e
        raise last

    def _numeric(self, A, MT):
        K, cur_mask, At = None, False, None
        for mask, Mr, plan in self.parts:
            if mask is not cur_mask:
                At = A.select_columns(mask) if mask is not None else A
                cur_mask = mask
            Kr = _ptap_numeric_plain(plan, At, Mr, MT, None, 1.0)
            K = Kr if K is None else K.add(Kr)
            del Kr
        return K

    def numeric(self, A, MT, zero_dofs=None, diag=1.0):
        K = self.__dict__.pop("_first", None)        # (the product formed while the plan was made, for the same A)
        if K is None or self._first_of is not A:
            K = self._numeric(A, MT)
        self._first_of = None
        if zero_dofs is not None and len(zero_dofs):
            K.zero_rows_cols(zero_dofs, diag, self.rows[2])
        return K


def _ptap_symbolic_plain(A, M, MT, a_row0=0, m_row0=0, mt_row0=0):
    h = handle()
    check(_lib.lib().tg_ptap_symbolic(A._h, int(a_row0), M._h, int(m_row0), MT._h, int(mt_row0), C.byref(h)),
          "tg_ptap_symbolic")
    plan = PtAPPlan(h)
    plan._rows = (int(a_row0), int(m_row0), int(mt_row0))
    return plan


def ptap_symbolic(A, M, MT, a_row0=0, m_row0=0, mt_row0=0):
    try:
        return _ptap_symbolic_plain(A, M, MT, a_row0, m_row0, mt_row0)
    except TigarHipError as e:
        if not _capacity_error(e):
            raise
        return SplitPtAPPlan(A, M, MT, a_row0, m_row0, mt_row0)


def ptap_prefer(kernels):
    """0: kernels of the general PtAP chosen from the operands, 1: wave-per-row Gustavson products, 2: fused
    workgroup-per-row kernel; returns the previous setting (``tg_ptap_prefer``)"""
    return int(_lib.lib().tg_ptap_prefer(int(kernels)))


def ptap_numeric(plan, A, M, MT, zero_dofs=None, diag=1.0):
    if isinstance(plan, SplitPtAPPlan):
        return plan.numeric(A, MT, zero_dofs, diag)
    try:
        return _ptap_numeric_plain(plan, A, M, MT, zero_dofs, diag)
    except TigarHipError as e:
        if not _capacity_error(e) or not hasattr(plan, "_rows"):
            raise
        # (the symbolic pass samples rows; a denser one shows up here)
        return SplitPtAPPlan(A, M, MT, *plan._rows).numeric(A, MT, zero_dofs, diag)


def _ptap_numeric_plain(plan, A, M, MT, zero_dofs=None, diag=1.0):
    h = handle()
    if zero_dofs is not None and len(zero_dofs):
        zd = _i32(zero_dofs)
        check(_lib.lib().tg_ptap_numeric(plan._h, A._h, M._h, MT._h, _p(zd, c_i32p), zd.size, float(diag),
                                         C.byref(h)), "tg_ptap_numeric")
    else:
        check(_lib.lib().tg_ptap_numeric(plan._h, A._h, M._h, MT._h, None, 0, float(diag), C.byref(h)),
              "tg_ptap_numeric")
    return DeviceCSR(h)


class FoldPlan(object):
    """K = R^T K_u R on stored places (``tg_foldplan_*``): ``create`` from a first product made with the general kernels,
    ``apply`` for every later K_u with the same pattern (None: another pattern -- use the general kernels)"""

    def __init__(self, h, keep):
        self._h, self._keep = h, keep

    @staticmethod
    def create(K_u, R, RT, ku_row0, rt_row0, K):
        h = handle()
        rc = _lib.lib().tg_foldplan_create(K_u._h, int(ku_row0), R._h, RT._h, int(rt_row0), K._h, C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_foldplan_create")
        return FoldPlan(h, (R, RT))

    def apply(self, K_u, zero_dofs=None, diag=1.0):
        zd = _i32(zero_dofs) if zero_dofs is not None and len(zero_dofs) else None
        out = handle()
        rc = _lib.lib().tg_foldplan_apply(self._h, K_u._h, _p(zd, c_i32p) if zd is not None else None,
                                          zd.size if zd is not None else 0, float(diag), C.byref(out))
        if rc == 100:
            return None
        check(rc, "tg_foldplan_apply")
        return DeviceCSR(out)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_foldplan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def ptap_kron(cur, cur_row0, dims_in, factors, out_row0, out_row1, zero_dofs=None, diag=1.0, intermediate=False,
              append_to=None):
    """One Kronecker contraction stage out = P^T cur P (dense-box kernel).  ``intermediate``: the
    result only feeds the next stage (or a vstack of such results) and is returned in the loose-row
    form (no row-reorder copy, no boundary conditions; see ``tg_ptap_kron_stage``).  ``append_to``: a
    ``CSRBuilder`` that receives the rows directly (``tg_ptap_kron_append``); returns True then.  ``factors[k]`` is a
    scipy CSR 1-D matrix (n_k x m_k) or None for the identity.  Returns None when the kernel
    declines (accumulator box too large for LDS) -- the caller then uses the general PtAP."""
    import scipy.sparse as sp
    d = len(dims_in)
    arr = (tg_kron1d_t * d)()
    keep = []
    for k in range(d):
        F = factors[k]
        if F is None:
            arr[k].n = int(dims_in[k])
            arr[k].m = int(dims_in[k])
            arr[k].rowptr = None
            continue
        F = sp.csr_matrix(F)
        F.sort_indices()
        FT = F.T.tocsr()
        FT.sort_indices()
        bufs = [_i32(F.indptr), _i32(F.indices), _f64(F.data), _i32(FT.indptr), _i32(FT.indices), _f64(FT.data)]
        keep += bufs
        arr[k].n, arr[k].m = F.shape
        arr[k].rowptr, arr[k].col, arr[k].val = _p(bufs[0], c_i32p), _p(bufs[1], c_i32p), _p(bufs[2], c_f64p)
        arr[k].t_rowptr, arr[k].t_col, arr[k].t_val = _p(bufs[3], c_i32p), _p(bufs[4], c_i32p), _p(bufs[5], c_f64p)
    dims = _i64(dims_in)
    h = handle()
    zd = _i32(zero_dofs) if zero_dofs is not None and len(zero_dofs) else None
    if append_to is not None:
        rc = _lib.lib().tg_ptap_kron_append(cur._h, int(cur_row0), d, _p(dims, c_i64p), arr, int(out_row0), int(out_row1),
                                            _p(zd, c_i32p) if zd is not None else None, zd.size if zd is not None else 0,
                                            float(diag), append_to._h)
        if rc == 100:
            return None
        check(rc, "tg_ptap_kron_append")
        return True
    if intermediate:
        if zd is not None:
            raise ValueError("boundary conditions belong to the last stage")
        rc = _lib.lib().tg_ptap_kron_stage(cur._h, int(cur_row0), d, _p(dims, c_i64p), arr, int(out_row0),
                                           int(out_row1), C.byref(h))
    else:
        rc = _lib.lib().tg_ptap_kron(cur._h, int(cur_row0), d, _p(dims, c_i64p), arr, int(out_row0), int(out_row1),
                                     _p(zd, c_i32p) if zd is not None else None, zd.size if zd is not None else 0,
                                     float(diag), C.byref(h))
    if rc == 100:
        return None
    check(rc, "tg_ptap_kron")
    return DeviceCSR(h)


# ------------------------------------------------------------------------------- Krylov
def krylov_solve(K, b, x, method="cg", pc="jacobi", rtol=1e-6, atol=1e-15, maxit=10000, restart=30, comm=None,
                 nonzero_initial_guess=False, stagnation_guard=False, symmetric=False):
    meth = {"cg": TG_KSP_CG, "gmres": TG_KSP_GMRES, "bicgstab": TG_KSP_BICGSTAB}[method]
    pcc = {"none": TG_PC_NONE, "jacobi": TG_PC_JACOBI, "chebyshev": TG_PC_CHEBYSHEV}[pc]
    iters, status, res = C.c_int(), C.c_int(), C.c_double()
    flags = (TG_KSP_NONZERO_GUESS if nonzero_initial_guess else 0) | (TG_KSP_STAGNATION_GUARD if stagnation_guard else 0) | \
        (TG_KSP_SYMMETRIC if symmetric else 0)
    check(_lib.lib().tg_krylov_solve_flags(K._h, b._h, x._h, meth, pcc, float(rtol), float(atol), int(maxit),
                                           int(restart), flags, comm._h if comm is not None else None,
                                           C.byref(iters), C.byref(res), C.byref(status)), "tg_krylov_solve")
    return iters.value, res.value, status.value


def lu_band_info(K):
    """(kl, ku, bytes of the band storage) of a square DeviceCSR"""
    kl, ku, nb = C.c_int(), C.c_int(), C.c_int64()
    check(_lib.lib().tg_lu_band_info(K._h, C.byref(kl), C.byref(ku), C.byref(nb)), "tg_lu_band_info")
    return kl.value, ku.value, nb.value


def lu_solve(K, b, x):
    """x = K^-1 b by banded LU with partial pivoting; returns LAPACK's info (0 = ok, j+1 = exact zero pivot at j)"""
    info = C.c_int()
    check(_lib.lib().tg_lu_solve(K._h, b._h, x._h, C.byref(info)), "tg_lu_solve")
    return info.value


def chol_solve(K, b, x):
    """x = K^-1 b by the blocked banded Cholesky factorisation (csrc/tg_chol.hip) if K is symmetric positive definite:
    True when it solved the system, False (x untouched) when K does not qualify"""
    done = C.c_int()
    check(_lib.lib().tg_chol_solve(K._h, b._h, x._h, C.byref(done)), "tg_chol_solve")
    return bool(done.value)


# ------------------------------------------------------------------------------- synthetic inputs
def kron_sum_csr(factors, row0=None, row1=None):
    """A = sum_t kron(F[t][d-1], ..., F[t][0]) (direction 0 fastest).  ``factors[t][k]`` are
    scipy CSR 1-D matrices; all terms must share one pattern per direction."""
    import scipy.sparse as sp
    nterms = len(factors)
    d = len(factors[0])
    arr = (tg_kron_dir_t * d)()
    keep = []
    total = 1
    for k in range(d):
        pat = sp.csr_matrix(factors[0][k])
        pat.sort_indices()
        n = pat.shape[0]
        total *= n
        vals = []
        for t in range(nterms):
            F = sp.csr_matrix(factors[t][k])
            F.sort_indices()
            if F.nnz != pat.nnz or not np.array_equal(F.indices, pat.indices) \
                    or not np.array_equal(F.indptr, pat.indptr):
                # bring onto the shared pattern (explicit zeros)
                F = (F + pat * 0.0).tocsr()
                F.sort_indices()
                if not np.array_equal(F.indices, pat.indices):
                    raise ValueError("1-D factors of direction %d do not share a pattern" % k)
            vals.append(_f64(F.data))
        rp, cl, vl = _i32(pat.indptr), _i32(pat.indices), _f64(np.concatenate(vals))
        keep += [rp, cl, vl]
        arr[k].n = n
        arr[k].rowptr = _p(rp, c_i32p)
        arr[k].col = _p(cl, c_i32p)
        arr[k].val = _p(vl, c_f64p)
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    h = handle()
    check(_lib.lib().tg_kron_sum_csr(d, nterms, arr, row0, row1, C.byref(h)), "tg_kron_sum_csr")
    return DeviceCSR(h)


def kron_csr_rect(factors, row0=None, row1=None):
    """kron(F[d-1], ..., F[0]) (direction 0 fastest) of rectangular scipy CSR 1-D factors, rows
    [row0,row1); explicit zeros of the factors are kept."""
    import scipy.sparse as sp
    d = len(factors)
    arr = (tg_kron_dir_t * d)()
    keep = []
    total = 1
    cdim = np.empty(d, dtype=np.int64)
    for k in range(d):
        F = sp.csr_matrix(factors[k])
        F.sort_indices()
        rp, cl, vl = _i32(F.indptr), _i32(F.indices), _f64(F.data)
        keep += [rp, cl, vl]
        arr[k].n = F.shape[0]
        arr[k].rowptr = _p(rp, c_i32p)
        arr[k].col = _p(cl, c_i32p)
        arr[k].val = _p(vl, c_f64p)
        total *= F.shape[0]
        cdim[k] = F.shape[1]
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    h = handle()
    check(_lib.lib().tg_kron_csr_rect(d, 1, arr, _p(cdim, c_i64p), row0, row1, 0, 0.0, 0, -1, C.byref(h)),
          "tg_kron_csr_rect")
    return DeviceCSR(h)


def kron3_csr(factors, row0=None, row1=None, col_offset=0, ncols_total=None):
    """kron(F[d-1], ..., F[0]) (direction 0 fastest) of rectangular scipy CSR 1-D factors WITHOUT explicit zeros, rows
    [row0,row1), values (v0*v1)*v2: the extraction matrix of a tensor B-spline (or its transpose, from the transposed
    factors) when the filter of generateM drops only exact zeros (``tg_kron3_csr``, pencil walk)."""
    import scipy.sparse as sp
    d = len(factors)
    arr = (tg_kron_dir_t * d)()
    keep = []
    total, ctotal = 1, 1
    cdim = np.empty(d, dtype=np.int64)
    for k in range(d):
        F = sp.csr_matrix(factors[k])
        F.sort_indices()
        rp, cl, vl = _i32(F.indptr), _i32(F.indices), _f64(F.data)
        keep += [rp, cl, vl]
        arr[k].n = F.shape[0]
        arr[k].rowptr = _p(rp, c_i32p)
        arr[k].col = _p(cl, c_i32p)
        arr[k].val = _p(vl, c_f64p)
        total *= F.shape[0]
        ctotal *= F.shape[1]
        cdim[k] = F.shape[1]
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    h = handle()
    check(_lib.lib().tg_kron3_csr(d, arr, _p(cdim, c_i64p), row0, row1, int(col_offset),
                                  int(ncols_total if ncols_total is not None else ctotal + col_offset), C.byref(h)),
          "tg_kron3_csr")
    return DeviceCSR(h)


def _patch(vertices, p, cp, nq):
    d = len(vertices)
    pt = tg_patch_t()
    keep = []
    pt.d, pt.p, pt.nsd, pt.nq = d, int(p), len(cp) - 1, int(nq)
    for k in range(d):
        v = _f64(vertices[k])
        keep.append(v)
        pt.verts[k] = _p(v, c_f64p)
        pt.nverts[k] = len(v)
    for c, vec in enumerate(cp):
        pt.cp[c] = vec._h
    return pt, keep


_MAPPED_FORMS = {"mass": 0, "laplace": 1, "biharmonic": 4}


def assemble_mapped_matrix(vertices, p, cp, form, nq=None, row0=None, row1=None, cp_node0=0):
    """FE mass (form 'mass'), stiffness ('laplace') or element-wise biharmonic ('biharmonic': int lap u lap v, nsd == d)
    matrix of the scalar Q_p space on the tensor
    grid with element ``vertices`` per direction, geometry F = cp[i]/cp[nsd] given by DeviceVectors on
    the FE nodes (dolfin.assemble stand-in, tIGAr/common.py:1206-1220, 917-945).  ``row0, row1``: the rows of
    whole node planes of the last direction only (global columns), ``cp`` then holding the nodes from ``cp_node0`` on."""
    pt, keep = _patch(vertices, p, cp, p + 1 if nq is None else nq)
    h = handle()
    if row0 is None and row1 is None and not cp_node0:
        check(_lib.lib().tg_assemble_mapped_matrix(C.byref(pt), _MAPPED_FORMS[form], C.byref(h)),
              "tg_assemble_mapped_matrix")
    else:
        check(_lib.lib().tg_assemble_mapped_matrix_rows(C.byref(pt), _MAPPED_FORMS[form], int(row0), int(row1),
                                                        int(cp_node0), C.byref(h)), "tg_assemble_mapped_matrix_rows")
    return DeviceCSR(h)


def assemble_mapped_elasticity_block(vertices, p, cp, i, j, lmbda, mu, nq=None, row0=None, row1=None, cp_node0=0):
    """Block (i, j) (test component i, trial component j) of a(u,v) = int lambda div u div v + 2 mu eps(u):eps(v) dx on the
    mapped patch (nsd == d), rows / window as ``assemble_mapped_matrix`` (``tg_assemble_mapped_elasticity_rows``)."""
    pt, keep = _patch(vertices, p, cp, p + 1 if nq is None else nq)
    h = handle()
    r0, r1 = (-1, -1) if (row0 is None and row1 is None) else (int(row0), int(row1))
    check(_lib.lib().tg_assemble_mapped_elasticity_rows(C.byref(pt), int(i), int(j), float(lmbda), float(mu), r0, r1,
                                                        int(cp_node0), C.byref(h)), "tg_assemble_mapped_elasticity_rows")
    return DeviceCSR(h)


def assemble_mapped_load(vertices, p, cp, fnodal, nq=None, row0=None, row1=None, cp_node0=0):
    """L(v) = int f_h v dx with f_h the nodal interpolant of the DeviceVector ``fnodal`` (on the nodes of ``cp``)."""
    pt, keep = _patch(vertices, p, cp, p + 1 if nq is None else nq)
    if row0 is None and row1 is None and not cp_node0:
        out = DeviceVector(n=fnodal.size())
        check(_lib.lib().tg_assemble_mapped_load(C.byref(pt), fnodal._h, out._h), "tg_assemble_mapped_load")
    else:
        out = DeviceVector(n=int(row1) - int(row0))
        check(_lib.lib().tg_assemble_mapped_load_rows(C.byref(pt), fnodal._h, int(row0), int(row1), int(cp_node0), out._h),
              "tg_assemble_mapped_load_rows")
    return out


def tensor_apply_1d(x, dims_in, k, F, col_shift=0, out=None):
    """Apply the scipy CSR 1-D factor ``F`` (rows = output indices of direction k, columns = input
    indices + col_shift) along direction ``k`` of the tensor-indexed DeviceVector ``x`` (direction 0
    fastest); returns the new DeviceVector (``tg_tensor_apply_1d``)."""
    import scipy.sparse as sp
    F = sp.csr_matrix(F)
    F.sort_indices()
    dims = _i64(dims_in)
    n_out = 1
    for j, n in enumerate(dims_in):
        n_out *= F.shape[0] if j == k else int(n)
    if out is None:
        out = DeviceVector(n=n_out, zero=False)          # (every entry is written by the pass)
    elif out.size() != n_out:
        raise ValueError("tensor_apply_1d: output vector has %d entries, expected %d" % (out.size(), n_out))
    rp, ci, fv = _i32(F.indptr), _i32(F.indices), _f64(F.data)
    check(_lib.lib().tg_tensor_apply_1d(len(dims_in), _p(dims, c_i64p), int(k), int(F.shape[0]), _p(rp, c_i32p),
                                        _p(ci, c_i32p), _p(fv, c_f64p), int(col_shift), x._h, out._h),
          "tg_tensor_apply_1d")
    return out


def vec_tensor3(b1d, scale=1.0, row0=None, row1=None):
    d = len(b1d)
    bs = [_f64(b) for b in b1d]
    n = _i64([b.shape[0] for b in bs])
    total = int(np.prod(n))
    row0 = 0 if row0 is None else int(row0)
    row1 = total if row1 is None else int(row1)
    out = DeviceVector(row1 - row0, zero=False)          # (every entry is written)
    ptrs = (c_f64p * d)(*[_p(b, c_f64p) for b in bs])
    check(_lib.lib().tg_vec_tensor3(out._h, d, ptrs, _p(n, c_i64p), float(scale), row0, row1), "tg_vec_tensor3")
    return out


def vec_copy_range(dst, dst_off, src, src_off, n):
    check(_lib.lib().tg_vec_copy_range(dst._h, int(dst_off), src._h, int(src_off), int(n)), "tg_vec_copy_range")


def vec_concat(parts):
    out = DeviceVector(sum(p.size() for p in parts))
    off = 0
    for p in parts:
        n = p.size()
        check(_lib.lib().tg_vec_copy_range(out._h, off, p._h, 0, n), "tg_vec_copy_range")
        off += n
    return out


# ------------------------------------------------------------------------------- timers / info
def timer_start(slot=0):
    check(_lib.lib().tg_timer_start(slot))


def timer_stop(slot=0):
    ms = C.c_double()
    check(_lib.lib().tg_timer_stop(slot, C.byref(ms)))
    return ms.value


def prof_reset():
    check(_lib.lib().tg_prof_reset())


def prof_get(slot=0):
    ms, n = C.c_double(), C.c_int64()
    check(_lib.lib().tg_prof_get(slot, C.byref(ms), C.byref(n)))
    return ms.value, n.value


def sync():
    check(_lib.lib().tg_sync())


def stream_set(i):
    """Make stream ``i`` (0 or 1) the current stream of the library (tg_stream_set)."""
    check(_lib.lib().tg_stream_set(int(i)), "tg_stream_set")


def stream_wait(waiter, waited):
    """Stream ``waiter`` waits (on the device) for everything enqueued so far on stream ``waited``."""
    check(_lib.lib().tg_stream_wait(int(waiter), int(waited)), "tg_stream_wait")


def device_count():
    """visible GPUs (no device is bound by this call)"""
    n = C.c_int()
    check(_lib.load(require_device=False).tg_device_count(C.byref(n)), "tg_device_count")
    return n.value


def device_info():
    name = C.create_string_buffer(256)
    ncu, hbm = C.c_int(), C.c_int64()
    check(_lib.lib().tg_device_info(name, 256, C.byref(ncu), C.byref(hbm)))
    return {"name": name.value.decode(), "num_cu": ncu.value, "hbm_bytes": hbm.value}


def pool_stats():
    """(bytes held free by the caching allocator, free blocks, blocks in use)"""
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    check(_lib.lib().tg_pool_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def mem_info():
    f, t = C.c_int64(), C.c_int64()
    check(_lib.lib().tg_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


# ------------------------------------------------------------------------------- multi-GPU
class Comm(object):
    """RCCL communicator (one process per GPU) + z-slab descriptor."""

    KINDS = ("rccl", "host", "ipc")

    def __init__(self, unique_id, rank, world, unique_id_halo=None):
        self._h = handle()
        self.rank, self.world = rank, world
        check(_lib.lib().tg_comm_create2(unique_id, unique_id_halo, rank, world, C.byref(self._h)), "tg_comm_create")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(_lib.lib().tg_comm_unique_id(buf), "tg_comm_unique_id")
        return buf.raw

    def set_slab(self, g0, g1, halo_lo, halo_hi, nglobal):
        self.g0, self.g1, self.halo_lo, self.halo_hi, self.nglobal = g0, g1, halo_lo, halo_hi, nglobal
        check(_lib.lib().tg_comm_set_slab(self._h, int(g0), int(g1), int(halo_lo), int(halo_hi), int(nglobal)),
              "tg_comm_set_slab")

    def halo_extend(self, x_local, xext=None):
        if xext is None:
            xext = DeviceVector(self.halo_lo + x_local.size() + self.halo_hi)
        check(_lib.lib().tg_comm_halo_extend(self._h, x_local._h, xext._h), "tg_comm_halo_extend")
        return xext

    def allreduce_sum(self, values):
        v = _f64(np.atleast_1d(values)).copy()
        check(_lib.lib().tg_comm_allreduce_sum(self._h, _p(v, c_f64p), v.size), "tg_comm_allreduce_sum")
        return v

    def info(self):
        """(rank, world, kind) as the communicator itself reports them (RCCL: ncclCommUserRank / ncclCommCount)"""
        r, w, k = C.c_int(), C.c_int(), C.c_int()
        check(_lib.lib().tg_comm_info(self._h, C.byref(r), C.byref(w), C.byref(k)), "tg_comm_info")
        return r.value, w.value, self.KINDS[k.value]

    def selftest(self, timeout_s=60.0):
        """small all-reduce + halo exchanges with known values, host waits bounded; raises on failure.  After a
        time-out the communicator must be abandoned, not destroyed (``abandon()``)."""
        rc = _lib.lib().tg_comm_selftest(self._h, float(timeout_s))
        if rc == 4:
            self.abandon()
        check(rc, "tg_comm_selftest")

    def abandon(self):
        """forget the handle without tearing the communicator down (its exchanges are stuck on the device)"""
        self._h = None

    def rank_devices(self):
        """device index of every rank as the ranks published them (IPC communicator; None entries otherwise)"""
        out = []
        for r in range(self.world):
            d = C.c_int(-1)
            check(_lib.lib().tg_comm_rank_device(self._h, r, C.byref(d)), "tg_comm_rank_device")
            out.append(d.value if d.value >= 0 else None)
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass


class HostComm(Comm):
    """Host-staged communicator (``tg_comm_create_host``): the solver's halo exchange and scalar reductions
    are staged through pinned host memory and carried by ``transport`` (``tigar_amd.launch.Transport``)."""

    def __init__(self, transport):
        self._h = handle()
        self.transport = transport
        self.rank, self.world = transport.rank, transport.world

        def allreduce(ctx, ptr, n):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(int(n),))
                transport.allreduce_sum(a)
                return 0
            except Exception as e:           # an exception must not unwind through the C frames
                self._err = e
                return 1

        def sendrecv(ctx, peer, sptr, ns, rptr, nr):
            try:
                send = np.ctypeslib.as_array(sptr, shape=(int(ns),)) if ns > 0 else np.zeros(0)
                recv = np.ctypeslib.as_array(rptr, shape=(int(nr),)) if nr > 0 else np.zeros(0)
                transport.sendrecv(int(peer), send, recv)
                return 0
            except Exception as e:
                self._err = e
                return 1
        self._err = None
        self._cb = (_lib.HOST_ALLREDUCE_FN(allreduce), _lib.HOST_SENDRECV_FN(sendrecv))   # keep the thunks alive
        check(_lib.lib().tg_comm_create_host(self.rank, self.world, self._cb[0], self._cb[1], None, C.byref(self._h)),
              "tg_comm_create_host")


class IpcComm(Comm):
    """IPC communicator (``tg_comm_create_ipc``): halo planes pushed into the neighbour's device mailbox through HIP
    IPC, flags and all-reduce slots in a shared-memory file, all waits inside kernels -- enqueue-only like RCCL, but
    it also serves ranks that share a GPU.  ``transport`` only carries the name of the shared file and two barriers."""

    def __init__(self, transport):
        import os
        import tempfile
        self._h = handle()
        self.transport = transport
        self.rank, self.world = transport.rank, transport.world
        nbytes = C.c_int64(0)
        check(_lib.lib().tg_comm_ipc_shm_bytes(C.byref(nbytes)), "tg_comm_ipc_shm_bytes")
        path = None
        if self.rank == 0:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            fd, path = tempfile.mkstemp(prefix="tigar_ipc_", dir=base)
            os.ftruncate(fd, nbytes.value)          # zero-filled
            os.close(fd)
        raw = (path or "").encode().ljust(512, b"\0")
        raw = transport.broadcast_bytes(raw, 512)
        path = raw.rstrip(b"\0").decode()
        err = None
        try:
            check(_lib.lib().tg_comm_create_ipc(path.encode(), self.rank, self.world, C.byref(self._h)),
                  "tg_comm_create_ipc")
        except Exception as e:                      # every rank has to reach the barrier
            err = e
        bad = np.array([0.0 if err is None else 1.0])
        transport.allreduce_sum(bad)
        if self.rank == 0:
            try:
                os.unlink(path)                     # the mappings keep the memory alive
            except OSError:
                pass
        if err is not None:
            raise err
        if bad[0] > 0.0:
            raise _lib.TigarHipError("IPC communicator: %d rank(s) could not set it up" % int(bad[0]))
