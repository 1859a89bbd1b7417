"""extractMatrix on a CONNECTED FE mesh with an extraction operator that is no Kronecker product: the cell-block product of
``cellptap`` after splitting the assembled matrix into one dense block per cell (csrc/tg_ptap_wave.hip, tg_elemsplit_*).

``A.PtAP(M)`` (tIGAr/common.py:1194-1195) for an A assembled by dolfin on ANY mesh is a sum over cells: every entry of A couples
two nodes of a common cell, so A = sum_c R_c^T A_c R_c for any assignment of its entries to cells holding both nodes (here: the
lowest such cell), and

    K = M^T A M = sum_c (R_c M)^T A_c (R_c M)

-- small dense triple products without any look-up (64 x 64 blocks for Q_3 in 3-D: 1.05 MFlop per cell instead of the 36 k
table look-ups per FE row of the row-wise product), merged into K by stored places.  What is needed besides the CSR operands is
the cells' node lists -- dolfin's ``V.dofmap().cell_dofs(c)``, here ``common._cell_dofs_arrays(grid)`` -- nothing about a
lattice.  The plan depends on M and the node lists (function list per cell, incidence, dense rows of M gathered on the device);
the splitting on the pattern of A (kept while the pattern stays).  VERDICT r4 #4.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import check, handle, c_i32p
from .device import DeviceCSR


class ElementSplitPtAP(object):
    def __init__(self, M, cellnodes):
        """``M``: DeviceCSR (FE rows x dofs); ``cellnodes``: [ncell, b] node numbers of every cell (b <= 64).  Raises ValueError
        when a cell names more than 64 functions or when the rows of M of a cell do not share one function list (the union of
        the functions of a cell must be the list of one of its nodes: true for spline spaces, whose functions are all non-zero
        in the cell's interior -- otherwise the general kernels are the right tool)."""
        cn = np.sort(np.asarray(cellnodes, dtype=np.int64), axis=1)
        ncell, b = cn.shape
        if not 1 <= b <= 64:
            raise ValueError("cells of %d nodes: beyond the 64 of the cell-block kernels" % b)
        nfe, ncp = M.shape
        indptr, indices = _pattern(M)
        rowlen = np.diff(indptr)
        lens = rowlen[cn]
        # the function list of a cell = the union of the functions of its nodes.  Spline spaces of degree >= 2 have nodes in
        # the cell's interior, where ALL functions of the cell are non-zero: then the longest row IS the union (checked: every
        # shorter row must be a subset -- the device fill of the dense rows finds a function that is not in the list).  p = 1
        # (every node holds one function) and anything else: the union proper, entry by entry.
        best = np.argmax(lens, axis=1)
        rbest = cn[np.arange(ncell), best]
        nf = lens[np.arange(ncell), best].astype(np.int32)
        total = int(lens.sum())
        quick = total <= 0 or int(nf.max()) * 2 > int(lens.mean() + 1)      # (one-entry rows: no row can be the union)
        if not quick or total < 50_000_000:
            rows = np.repeat(cn.ravel(), lens.ravel())
            start = np.repeat(indptr[cn.ravel()], lens.ravel())
            within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(lens.ravel()) - lens.ravel(), lens.ravel())
            cols = indices[start + within].astype(np.int64)
            cell_of = np.repeat(np.repeat(np.arange(ncell, dtype=np.int64), b), lens.ravel())
            ukey = np.unique(cell_of * ncp + cols)
            ucell, ufun0 = ukey // ncp, ukey % ncp
            first = np.searchsorted(ucell, np.arange(ncell + 1, dtype=np.int64))
            nf = np.diff(first).astype(np.int32)
            nfmax = int(nf.max())
            if nfmax > 64 or nfmax < 1:
                raise ValueError("a cell names %d functions: beyond the 64 of the cell-block kernels" % nfmax)
            q = np.arange(nfmax, dtype=np.int64)[None, :]
            used = q < nf[:, None]
            fl = np.zeros((ncell, nfmax), dtype=np.int32)
            fl[ucell, np.arange(ukey.size, dtype=np.int64) - first[ucell]] = ufun0
            del rows, start, within, cols, cell_of
        else:
            nfmax = int(nf.max())
            if nfmax > 64 or nfmax < 1:
                raise ValueError("a cell names %d functions: beyond the 64 of the cell-block kernels" % nfmax)
            q = np.arange(nfmax, dtype=np.int64)[None, :]
            used = q < nf[:, None]
            flat = indptr[rbest][:, None] + q
            fl = np.zeros((ncell, nfmax), dtype=np.int32)
            fl[used] = indices[flat[used]]
        # incidence: dof i -> rows (c, q) of the element matrices that hold it
        cidx = np.broadcast_to(np.arange(ncell, dtype=np.int64)[:, None], (ncell, nfmax))[used]
        ufun = fl[used].astype(np.int64)
        erow = (cidx * nfmax + np.broadcast_to(q, (ncell, nfmax))[used])
        order = np.argsort(ufun, kind="stable")
        inc = sp.csr_matrix((np.ones(ufun.size), erow[order], np.concatenate([[0], np.cumsum(np.bincount(ufun, minlength=ncp))])),
                            shape=(ncp, ncell * nfmax))
        Xb = sp.csr_matrix((np.ones(ufun.size, dtype=np.int8), (ufun, cidx)), shape=(ncp, ncell))
        klen = np.diff((Xb @ Xb.T).tocsr().indptr) if ncp else np.zeros(0, dtype=np.int64)
        # node -> cells (ascending), for the ownership rule of the splitting
        nodes = cn.ravel()
        cells = np.repeat(np.arange(ncell, dtype=np.int64), b)
        o2 = np.argsort(nodes, kind="stable")
        self._ncells = np.ascontiguousarray(cells[o2], dtype=np.int32)
        self._nptr = np.ascontiguousarray(np.concatenate([[0], np.cumsum(np.bincount(nodes, minlength=nfe))]), dtype=np.int32)
        self._cn = np.ascontiguousarray(cn, dtype=np.int32)
        self.shape = (ncp, ncp)
        self.b, self.ncell, self.nfmax, self.nfe = int(b), int(ncell), nfmax, int(nfe)
        self._inc = DeviceCSR.from_scipy(inc)            # (borrowed by the plan: kept alive here)
        self._h = handle()
        fl = np.ascontiguousarray(fl)
        nfc = np.ascontiguousarray(nf)
        check(_lib.lib().tg_cellplan_create_from_rows(ncell, int(b), nfmax, M._h, self._cn.ctypes.data_as(c_i32p),
                                                      fl.ctypes.data_as(c_i32p), nfc.ctypes.data_as(c_i32p), self._inc._h,
                                                      int(klen.max()) if klen.size else 1,
                                                      float(klen.mean()) if klen.size else 1.0, C.byref(self._h)),
              "tg_cellplan_create_from_rows")
        self._split, self._split_key = None, None

    def _splitting(self, A):
        key = (A.shape, A.nnz)
        if self._split is not None and self._split_key == key:
            return self._split
        self._drop_split()
        h = handle()
        rc = _lib.lib().tg_elemsplit_create(A._h, self.ncell, self.b, self._cn.ctypes.data_as(c_i32p),
                                            self._nptr.ctypes.data_as(c_i32p), self._ncells.ctypes.data_as(c_i32p), C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_elemsplit_create")
        self._split, self._split_key = h, key
        return h

    def ptap(self, A, zero_dofs=None, diag=1.0):
        """M^T A M with MatZeroRowsColumns fused, or None when A holds an entry whose nodes share no cell"""
        if A.shape != (self.nfe, self.nfe):
            return None
        zd = np.ascontiguousarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        for attempt in (0, 1):
            s = self._splitting(A)
            if s is None:
                return None
            h = handle()
            rc = _lib.lib().tg_elemsplit_ptap(s, self._h, A._h, zd.ctypes.data_as(c_i32p) if zd.size else None, zd.size,
                                              float(diag), C.byref(h))
            if rc == 100 and attempt == 0:
                self._drop_split()            # same size, another pattern: split again (once)
                continue
            check(rc, "tg_elemsplit_ptap")
            return DeviceCSR(h)
        return None

    def _drop_split(self):
        if getattr(self, "_split", None):
            _lib.lib().tg_elemsplit_destroy(self._split)
        self._split, self._split_key = None, None

    def __del__(self):
        try:
            self._drop_split()
            if self._h:
                _lib.lib().tg_cellplan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _pattern(M):
    """(indptr int64, indices int32) of a DeviceCSR on the host"""
    S = M.to_scipy().tocsr()
    S.sort_indices()
    return S.indptr.astype(np.int64), S.indices
