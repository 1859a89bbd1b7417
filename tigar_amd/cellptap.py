"""extractMatrix on cell-local FE spaces: K = sum_c S_c^T (M_c^T A_c M_c) S_c  (csrc/tg_ptap_wave.hip, cell-block product).

The reference extracts T-splines and multi-patch B-splines to meshes of DISCONNECTED cells (tIGAr/RhinoTSplines.py:195-240,
tIGAr/BSplines.py:800-860): every cell carries its own (p+1)^d nodes, numbered cell after cell, so whatever dolfin assembles
on such a space is block diagonal with one dense block per cell, and the rows of M that belong to a cell name a short list
of spline functions.  MatPtAP (tIGAr/common.py:1194-1195) then is a sum of small dense triple products -- the multiply-adds
need no look-up at all, only the entries of an element matrix are merged into K by look-up.

The PLAN depends on M alone (the symbolic half of the product): per cell the sorted list of its functions, its rows of M as a
dense block over that list, and for every function the rows (cell, position) of the element matrices that hold it.  Built
here with numpy from the stored pattern of M (once per extraction operator, cached on the spline), kept on the device by
``tg_cellplan_create``.  ``ptap`` verifies on the device that A is block diagonal with dense b x b blocks and returns None
otherwise (the caller then uses the general kernels).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, handle, c_f64p, c_i32p
from .device import DeviceCSR


def block_size_of(A):
    """b if the DeviceCSR ``A`` can be block diagonal with dense b x b blocks (entry count and shape only; the device check of
    ``ptap`` decides), else 0"""
    n = A.shape[0]
    if n == 0 or A.shape[1] != n or A.nnz % n:
        return 0
    b = A.nnz // n
    return int(b) if 2 <= b <= 64 and n % b == 0 else 0


def cell_size_with_extras(A):
    """b if ``A`` can be a block-diagonal matrix with dense b x b cell blocks PLUS a few couplings outside the blocks (fewer
    than one per row on average: contact or penalty terms added by hand, demos/kl-shell-svk/reef-knot.py:455-467), else 0"""
    n = A.shape[0]
    if n == 0 or A.shape[1] != n:
        return 0
    b = A.nnz // n
    return int(b) if 2 <= b <= 64 and n % b == 0 and A.nnz > n * b else 0


def split_cells(A, b):
    """(D, R) with A = D + R: the dense diagonal cell blocks and the remainder (``tg_csr_split_cells``), or None when some
    row lacks entries of its own block"""
    d, r = handle(), handle()
    rc = _lib.lib().tg_csr_split_cells(A._h, int(b), C.byref(d), C.byref(r))
    if rc == 100:
        return None
    check(rc, "tg_csr_split_cells")
    return DeviceCSR(d), DeviceCSR(r)


def nonempty_rows(R):
    """ascending indices of the rows of the DeviceCSR ``R`` that hold entries (``tg_csr_nonempty_rows``)"""
    from ._lib import c_i64p
    n = C.c_int64()
    check(_lib.lib().tg_csr_nonempty_rows(R._h, 0, None, C.byref(n)), "tg_csr_nonempty_rows")
    rows = np.zeros(max(1, n.value), dtype=np.int64)
    if n.value:
        check(_lib.lib().tg_csr_nonempty_rows(R._h, n.value, rows.ctypes.data_as(c_i64p), C.byref(n)), "tg_csr_nonempty_rows")
    return rows[:n.value]


def remainder_product(R, M, cache=None):
    """M^T R M for a matrix R with few non-empty rows (the couplings outside the cell blocks): the general kernels on the
    operands restricted to those rows -- M^T's columns and R's rows compacted to the rows that hold anything -- instead of
    a pass over every row of M^T.  ``cache`` (a dict): the symbolic plan is kept while the rows and the entry count of R
    stay what they were (penalty terms that persist over the steps of a Newton loop); a first product on a new pattern
    costs ~8 ms at 4 M FE rows (the plan's discovering pass), a repeated one ~2.5 ms.  None when R is empty."""
    from . import device as _dev
    rows = nonempty_rows(R)
    if rows.size == 0:
        return None
    Rs = R.gather_rows(rows)                     # (few rows, global columns)
    key = (R.shape, R.nnz, rows.size, hash(rows.tobytes()))
    if cache is not None and cache.get("key") == key:
        try:
            return _dev.ptap_numeric(cache["plan"], Rs, M, cache["MTs"])
        except _dev.TigarHipError:               # (same rows and count, other columns: plan again)
            pass
    MTs = M.gather_rows(rows).transpose()        # dofs x those rows
    plan = _dev.ptap_symbolic(Rs, M, MTs)
    K = _dev.ptap_numeric(plan, Rs, M, MTs)
    if cache is not None:
        cache.update(key=key, plan=plan, MTs=MTs)
    return K


class CellBlockPtAP(object):
    def __init__(self, M, b):
        """``M``: the extraction operator (DeviceCSR, FE rows x dofs), ``b``: nodes per cell.  Raises ValueError when the
        cells' function lists are longer than 64 (then the general kernels are the right tool)."""
        Ms = M.to_scipy().tocsr()
        Ms.sort_indices()
        nfe, ncp = Ms.shape
        if nfe % b:
            raise ValueError("the FE rows are not a whole number of cells")
        ncell = nfe // b
        rows = np.repeat(np.arange(nfe, dtype=np.int64), np.diff(Ms.indptr))
        cell = rows // b
        key = cell * ncp + Ms.indices.astype(np.int64)                 # (cell, function) of every stored entry
        ukey, inv = np.unique(key, return_inverse=True)                 # sorted: by cell, then by function
        ucell, ufun = ukey // ncp, (ukey % ncp).astype(np.int32)
        first = np.searchsorted(ucell, np.arange(ncell + 1, dtype=np.int64))
        nf = np.diff(first).astype(np.int32)
        nfmax = int(nf.max()) if ncell else 0
        if nfmax > 64 or nfmax < 1:
            raise ValueError("a cell names %d functions: beyond the 64 of the cell-block kernels" % nfmax)
        upos = (np.arange(ukey.size, dtype=np.int64) - first[ucell]).astype(np.int64)      # position in its cell's list
        fl = np.zeros((ncell, nfmax), dtype=np.int32)
        fl[ucell, upos] = ufun
        md = np.zeros((ncell, b, nfmax), dtype=np.float64)
        md[cell, rows - cell * b, upos[inv]] = Ms.data
        # incidence: dof i -> rows (c, q) of the element matrices
        erow = ucell * nfmax + upos
        order = np.argsort(ufun, kind="stable")
        import scipy.sparse as sp
        inc = sp.csr_matrix((np.ones(ukey.size), erow[order].astype(np.int64), np.concatenate(
            [[0], np.cumsum(np.bincount(ufun, minlength=ncp))])), shape=(ncp, ncell * nfmax))
        # row lengths of K: functions that share a cell
        Xb = sp.csr_matrix((np.ones(ukey.size, dtype=np.int8), (ufun.astype(np.int64), ucell)), shape=(ncp, ncell))
        klen = np.diff((Xb @ Xb.T).tocsr().indptr) if ncp else np.zeros(0, dtype=np.int64)
        self.shape = (ncp, ncp)
        self.b, self.ncell, self.nfmax = int(b), int(ncell), nfmax
        self._inc = DeviceCSR.from_scipy(inc)            # (borrowed by the plan: kept alive here)
        self._h = handle()
        nfc = np.ascontiguousarray(nf)
        check(_lib.lib().tg_cellplan_create(ncell, int(b), nfmax, ncp, md.ctypes.data_as(c_f64p), fl.ctypes.data_as(c_i32p),
                                            nfc.ctypes.data_as(c_i32p), self._inc._h, int(klen.max()) if klen.size else 1,
                                            float(klen.mean()) if klen.size else 1.0, C.byref(self._h)), "tg_cellplan_create")

    def ptap(self, A, zero_dofs=None, diag=1.0):
        """M^T A M with MatZeroRowsColumns fused, or None when ``A`` is not block diagonal with dense b x b blocks"""
        h = handle()
        zd = np.ascontiguousarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        rc = _lib.lib().tg_cellplan_ptap(self._h, A._h, zd.ctypes.data_as(c_i32p) if zd.size else None, zd.size, float(diag),
                                         C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_cellplan_ptap")
        return DeviceCSR(h)

    def ptap_extras(self, A):
        """(M^T D M, R) for A = D + R, D the dense cell blocks read in place, R the couplings outside them (A's shape); None
        when some row lacks entries of its own block (``tg_cellplan_ptap_extras``)"""
        k, r = handle(), handle()
        rc = _lib.lib().tg_cellplan_ptap_extras(self._h, A._h, C.byref(k), C.byref(r))
        if rc == 100:
            return None
        check(rc, "tg_cellplan_ptap_extras")
        return DeviceCSR(k), DeviceCSR(r)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_cellplan_destroy(self._h)
                self._h = None
        except Exception:
            pass
