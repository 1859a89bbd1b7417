"""extractMatrix on a CONNECTED FE mesh with operands used as general sparse matrices: the element split (csrc/tg_elemsplit.hip).

``A.PtAP(M)`` (tIGAr/common.py:1194-1195) for an A assembled by dolfin on ANY mesh is a sum over cells: every entry of A couples
two nodes of a common cell, so A = sum_c R_c^T A_c R_c for any assignment of its entries to cells holding both nodes (here: the
first such cell), and

    K = M^T A M = sum_c (R_c M)^T A_c (R_c M)

-- small dense triple products without any look-up (64 x 64 blocks for Q_3 in 3-D: 1.05 MFlop per cell instead of the 36 k
table look-ups per FE row of the row-wise product), merged into K by stored places.  What is needed besides the CSR operands is
the cells' node lists -- dolfin's ``V.dofmap().cell_dofs(c)``, here ``CellNodes`` -- nothing about a lattice.

Round 6: everything on the device (the plan of round 5 was numpy / scipy on a downloaded M: 2.7 s at 64^3 p = 3), and in CHUNKS
of cells whose products are added -- the form in which the streamed / multi-rank engine (``dist.SlabHotPath``) uses it for
operands that do not fit the device or are spread over ranks (``ElementChunkAdder``).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, handle, c_i32p, c_i64p
from .device import DeviceCSR


class CellNodes(object):
    """node lists of the cells of an FE mesh on the device ([ncell][b] int32, b <= 128)"""

    def __init__(self, h):
        self._h = h
        n, b = C.c_int64(), C.c_int()
        check(_lib.lib().tg_cells_dims(h, C.byref(n), C.byref(b)), "tg_cells_dims")
        self.ncell, self.b = int(n.value), int(b.value)

    @staticmethod
    def from_host(cellnodes):
        cn = np.ascontiguousarray(cellnodes, dtype=np.int32)
        if cn.ndim != 2:
            raise ValueError("cell node lists: a [ncell, nodes per cell] array is expected")
        if not 1 <= cn.shape[1] <= 128:
            raise ValueError("cells of %d nodes: beyond the 128 of the element kernels" % cn.shape[1])
        h = handle()
        check(_lib.lib().tg_cells_from_host(cn.ctypes.data_as(c_i32p), cn.shape[0], cn.shape[1], C.byref(h)), "tg_cells_from_host")
        return CellNodes(h)

    @staticmethod
    def from_grid(grid, elem_lo=None, elem_hi=None):
        """the cells of a continuous ``TensorNodeGrid`` (this package's stand-in of dolfin's Q_p space: ``common._cell_dofs_arrays``
        in the same order), all of them or the box ``elem_lo[k] <= e_k < elem_hi[k]``, generated on the device"""
        d, p = grid.dim(), int(grid.degree)
        if getattr(grid, "dg", False) or p < 1:
            raise ValueError("cells from a grid: a continuous space of degree >= 1 is expected")
        nn = np.ascontiguousarray(grid.shape(), dtype=np.int64)
        nel = (nn - 1) // p
        lo = np.zeros(d, dtype=np.int64) if elem_lo is None else np.ascontiguousarray(elem_lo, dtype=np.int64)
        hi = nel.astype(np.int64) if elem_hi is None else np.ascontiguousarray(elem_hi, dtype=np.int64)
        h = handle()
        check(_lib.lib().tg_cells_from_grid(d, nn.ctypes.data_as(c_i64p), p, lo.ctypes.data_as(c_i64p), hi.ctypes.data_as(c_i64p),
                                            C.byref(h)), "tg_cells_from_grid")
        return CellNodes(h)

    def to_host(self):
        out = np.empty((self.ncell, self.b), dtype=np.int32)
        check(_lib.lib().tg_cells_download(self._h, out.ctypes.data_as(c_i32p)), "tg_cells_download")
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_cells_destroy(self._h)
                self._h = None
        except Exception:
            pass


class ElementChunk(object):
    """the plan of the product over the cells [own0, own1) of ``cells`` (the others only take part in the ownership rule);
    ``M``: DeviceCSR holding the rows [m_row0, m_row0 + M.shape[0]) of the extraction operator with global columns.  Raises
    ``ValueError`` when the cells do not qualify (status 100 of the library)."""

    def __init__(self, cells, M, m_row0=0, own=None):
        self.cells, self.M, self.m_row0 = cells, M, int(m_row0)
        own0, own1 = (0, cells.ncell) if own is None else (int(own[0]), int(own[1]))
        self._h = handle()
        rc = _lib.lib().tg_elemplan_create(cells._h, own0, own1, M._h, self.m_row0, C.byref(self._h))
        if rc == 100:
            self._h = None
            raise ValueError("the cells do not qualify for the element split (functions per cell / cells per function > 128)")
        check(rc, "tg_elemplan_create")
        d0, d1, nk = C.c_int64(), C.c_int64(), C.c_int64()
        nf, ni = C.c_int(), C.c_int()
        check(_lib.lib().tg_elemplan_info(self._h, C.byref(d0), C.byref(d1), C.byref(nf), C.byref(ni), C.byref(nk)), "tg_elemplan_info")
        self.dofs = (int(d0.value), int(d1.value))          # the rows of K this chunk contributes to
        self.nfmax, self.ninc_max = int(nf.value), int(ni.value)

    def ptap(self, A, a_row0=0, check_rows=None, zero_dofs=None, diag=1.0):
        """rows ``self.dofs`` of sum_{own cells} (R_c M)^T A_c (R_c M) as a DeviceCSR with global columns, or None when an entry of
        the rows ``check_rows`` of A (default: all rows handed in) couples two nodes without a listed common cell"""
        c0, c1 = (a_row0, a_row0 + A.shape[0]) if check_rows is None else check_rows
        zd = np.ascontiguousarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        h = handle()
        rc = _lib.lib().tg_elemplan_ptap(self._h, A._h, int(a_row0), int(c0), int(c1),
                                         zd.ctypes.data_as(c_i32p) if zd.size else None, zd.size, float(diag), C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_elemplan_ptap")
        return DeviceCSR(h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().tg_elemplan_destroy(self._h)
                self._h = None
        except Exception:
            pass


class ElementSplitPtAP(object):
    """the whole mesh in one chunk: M^T A M for a resident M and A (``ExtractedSpline.extractMatrix``)"""

    def __init__(self, M, cellnodes):
        """``M``: DeviceCSR (FE rows x dofs); ``cellnodes``: ``CellNodes`` or a host array [ncell, b]"""
        cells = cellnodes if isinstance(cellnodes, CellNodes) else CellNodes.from_host(cellnodes)
        self.shape = (M.shape[1], M.shape[1])
        self.nfe = M.shape[0]
        self.ncell, self.b = cells.ncell, cells.b
        self._chunk = ElementChunk(cells, M)
        self.nfmax = self._chunk.nfmax

    def ptap(self, A, zero_dofs=None, diag=1.0):
        """M^T A M with MatZeroRowsColumns fused, or None when A holds an entry whose nodes share no cell"""
        if A.shape != (self.nfe, self.nfe):
            return None
        K = self._chunk.ptap(A, 0, None, zero_dofs, diag)
        if K is None:
            return None
        d0, d1 = self._chunk.dofs
        if (d0, d1) != (0, self.shape[0]):           # (functions before the first / after the last one the cells hold: empty rows)
            K = _pad_rows(K, d0, self.shape[0])
            if zero_dofs is not None and len(zero_dofs):
                K.zero_rows_cols(np.asarray(zero_dofs, dtype=np.int32), float(diag))
        return K


def _pad_rows(K, row0, nrows_total):
    """K as the rows [row0, row0 + K.shape[0]) of a matrix of nrows_total rows (the others empty)"""
    import scipy.sparse as sp
    from .device import csr_vstack
    ncols = K.shape[1]
    parts = []
    if row0 > 0:
        parts.append(DeviceCSR.from_scipy(sp.csr_matrix((row0, ncols))))
    parts.append(K)
    if row0 + K.shape[0] < nrows_total:
        parts.append(DeviceCSR.from_scipy(sp.csr_matrix((nrows_total - row0 - K.shape[0], ncols))))
    return csr_vstack(parts) if len(parts) > 1 else K
