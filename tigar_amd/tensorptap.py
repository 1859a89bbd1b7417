"""
Host side of the tensor-pattern extractMatrix path (csrc/tg_tensor_body.h, csrc/tg_ptap_tensor.hip):
K = M^T A M (tIGAr/common.py:1176-1204) for a tensor-product B-spline patch whose FE matrix carries the
element-coupling pattern of the Q_p node grid, in three line-walk passes without column decode, LDS or atomics.

This module decides -- from the 1-D extraction matrices alone -- whether a patch has the structure the walk
relies on, and builds the per-element local weight tables.  The FE matrix itself is checked on the device,
entry by entry, while it is read (``planes`` returns None when it has another pattern and the caller falls
back to the general kernels).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import device as _dev
from ._lib import check, handle, c_f64p, c_i32p, tg_tensor_dir_t, tg_tensor_pair_dir_t


# Structure checks and plans are functions of the 1-D extraction matrices alone; generators are rebuilt often (every
# pass of a Newton loop or a benchmark step builds a new one on the same knot vectors), so both are kept keyed on the
# CONTENT of those matrices (digest of values and indices) -- never on object identity.
_WL_CACHE = {}
_PLAN_CACHE = {}
_CACHE_MAX = 16


def _digest(M1):
    import hashlib
    M1 = M1.tocsr()
    h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(M1.indptr).tobytes())
    h.update(np.ascontiguousarray(M1.indices).tobytes())
    h.update(np.ascontiguousarray(M1.data).tobytes())
    return h.digest()


def checked_weights(M1, p, nel):
    """``local_weights`` when the direction also passes ``band_pattern_ok``, else None; cached on the content of M1"""
    key = (int(p), int(nel), M1.shape, _digest(M1))
    if key not in _WL_CACHE:
        if len(_WL_CACHE) >= 4 * _CACHE_MAX:
            _WL_CACHE.clear()
        wl = local_weights(M1, p, nel)
        _WL_CACHE[key] = wl if wl is not None and band_pattern_ok(M1, p, nel) else None
    return _WL_CACHE[key], key


def _cached_plan(kind, p, nels, nfields, keys, build):
    key = (kind, int(p), tuple(int(n) for n in nels), int(nfields), tuple(keys))
    plan = _PLAN_CACHE.pop(key, None)
    if plan is None:
        plan = build()
        while len(_PLAN_CACHE) >= _CACHE_MAX:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    _PLAN_CACHE[key] = plan                     # (most recently used last)
    return plan


def local_weights(M1, p, nel):
    """wl[e, j, q] = M1[p*e + j, e + q] (value at node j of element e of spline function e+q), or None when a
    stored entry of M1 lies outside that window, i.e. the direction does not have the structure of an open
    knot vector with simple interior knots on the CG degree-p grid."""
    M1 = M1.tocsr()
    nfe, ncp = M1.shape
    if nfe != p * nel + 1 or ncp != nel + p:
        return None
    a = np.repeat(np.arange(nfe), np.diff(M1.indptr))
    c, v = M1.indices.astype(np.int64), M1.data
    keep = v != 0.0
    a, c, v = a[keep], c[keep], v[keep]
    wl = np.zeros((nel, p + 1, p + 1))
    vertex = (a % p) == 0
    # every node belongs to element a // p as local node a % p (not the last node) ...
    m1 = a < nfe - 1
    e1, j1 = a[m1] // p, a[m1] % p
    q1 = c[m1] - e1
    # ... and a vertex node (but the first) also to the element before it, as its local node p
    m2 = vertex & (a > 0)
    e2 = a[m2] // p - 1
    q2 = c[m2] - e2
    if np.any(q1 < 0) or np.any(q1 > p) or np.any(q2 < 0) or np.any(q2 > p):
        return None
    wl[e1, j1, q1] = v[m1]
    wl[e2, p, q2] = v[m2]
    return wl


def local_weights_padded(M1, P, nel, ps):
    """``local_weights`` for a spline of degree ``ps`` <= P extracted to the CG degree-P grid (a component of a compatible
    spline, tIGAr/compatibleSplines.py:21-66: FE degree = the largest directional degree of the field): wl[e, j, q] =
    M1[P*e + j, e + q] for q <= ps, zero for the P - ps functions the padding adds.  None when a stored entry lies outside
    [e, e + ps] or a function e + q is stored at no node of element e (the product's band would not be full)."""
    M1 = M1.tocsr()
    nfe, ncp = M1.shape
    if ps < 1 or ps > P or nfe != P * nel + 1 or ncp != nel + ps:
        return None
    a = np.repeat(np.arange(nfe), np.diff(M1.indptr))
    c, v = M1.indices.astype(np.int64), M1.data
    keep = v != 0.0
    a, c, v = a[keep], c[keep], v[keep]
    wl = np.zeros((nel, P + 1, P + 1))
    m1 = a < nfe - 1
    e1, j1 = a[m1] // P, a[m1] % P
    q1 = c[m1] - e1
    m2 = ((a % P) == 0) & (a > 0)
    e2 = a[m2] // P - 1
    q2 = c[m2] - e2
    if np.any(q1 < 0) or np.any(q1 > ps) or np.any(q2 < 0) or np.any(q2 > ps):
        return None
    wl[e1, j1, q1] = v[m1]
    wl[e2, P, q2] = v[m2]
    if not (wl[:, :, :ps + 1] != 0.0).any(axis=1).all():
        return None
    # (the closing vertex of an element carries no weight of the element's first function: the walk relies on it)
    if nel > 1 and np.any(wl[:-1, P, 0] != 0.0):
        return None
    return wl


def band_pattern_ok(M1, p, nel):
    """structural pattern of the 1-D K (= M1^T pattern(A1) M1 as PETSc's symbolic product sees it) is the
    full band |i - i'| <= p clipped to the matrix: with every stored column of element e's nodes inside
    [e, e+p] (``local_weights``) the band cannot be exceeded, and it is full iff every pair of functions
    (e+q, e+q') is stored at some node of element e, for every e."""
    wl = local_weights(M1, p, nel)
    if wl is None:
        return False
    present = (wl != 0.0).any(axis=1)                 # [nel, p+1]: function e+q is stored at some node of element e
    return bool(present.all())


class TensorPlanes(object):
    """B2 planes of FE planes [z0, z1) (device, dense blocks); input of the z pass"""

    def __init__(self, h, z0, z1):
        self._h, self.z0, self.z1 = h, z0, z1

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_tensor_planes_destroy(self._h)
                self._h = None
        except Exception:
            pass


class TensorPtAP(object):
    """Plan of the tensor-pattern PtAP for one patch (1-D tables in HBM)."""

    def __init__(self, p, nels, wls, pair=None):
        """``pair``: (row-side weights, row degrees, column degrees) per direction for a block with different spline bases
        on its two sides (``wls`` then being the column side's, padded) -- ``tg_tensor_plan_create_pair``"""
        self.p, self.nels = int(p), [int(n) for n in nels]
        self._keep = [np.ascontiguousarray(w, dtype=np.float64) for w in wls]
        self._h = handle()
        if pair is None:
            self.pr = self.pc = [self.p] * 3
            arr = (tg_tensor_dir_t * 3)()
            for k in range(3):
                arr[k].p = self.p
                arr[k].nel = self.nels[k]
                arr[k].wl = self._keep[k].ctypes.data_as(c_f64p)
            check(_lib.lib().tg_tensor_plan_create(3, arr, C.byref(self._h)), "tg_tensor_plan_create")
            return
        wlr, self.pr, self.pc = pair
        self.pr, self.pc = [int(v) for v in self.pr], [int(v) for v in self.pc]
        self._keep_r = [np.ascontiguousarray(w, dtype=np.float64) for w in wlr]
        arr = (tg_tensor_pair_dir_t * 3)()
        for k in range(3):
            arr[k].p = self.p
            arr[k].nel = self.nels[k]
            arr[k].pr, arr[k].pc = self.pr[k], self.pc[k]
            arr[k].wlr = self._keep_r[k].ctypes.data_as(c_f64p)
            arr[k].wlc = self._keep[k].ctypes.data_as(c_f64p)
        check(_lib.lib().tg_tensor_plan_create_pair(3, arr, C.byref(self._h)), "tg_tensor_plan_create_pair")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_tensor_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def structure(kx):
        """(p, nels, wls) if the patch of the ``KronExtraction`` has the structure of the fast path, else None"""
        if kx.d != 3:
            return None
        grid = kx.grid
        ps = [s.p for s in kx.basis.splines]
        p = ps[0]
        if any(q != p for q in ps) or p < 1 or p > 3 or getattr(grid, "dg", False) or grid.degree != p:
            return None
        nels, wls, keys = [], [], []
        for k in range(3):
            nel = len(grid.vertices[k]) - 1
            wl, key = checked_weights(kx.M1[k], p, nel)
            if wl is None:
                return None
            nels.append(nel)
            wls.append(wl)
            keys.append(key)
        kx._tensor_keys = keys
        return p, nels, wls

    @staticmethod
    def for_extraction(kx):
        """plan cached on the ``KronExtraction`` (None if the patch does not qualify or TIGAR_PTAP_TENSOR=0); patches
        with the same 1-D extraction matrices share one plan"""
        if os.environ.get("TIGAR_PTAP_TENSOR", "1") == "0":
            return None
        if not hasattr(kx, "_tensor_plan"):
            st = TensorPtAP.structure(kx)
            kx._tensor_plan = _cached_plan("3d", st[0], st[1], 1, kx._tensor_keys, lambda: TensorPtAP(*st)) \
                if st is not None else None
        return kx._tensor_plan

    @staticmethod
    def structure_pair(kx_row, kx_col):
        """(P, nels, column weights, (row weights, row degrees, column degrees), keys) when block (row basis, column basis)
        of a space on ONE Q_P node grid has the structure of the walks, else None"""
        if kx_row.d != 3 or kx_col.d != 3:
            return None
        g, g2 = kx_row.grid, kx_col.grid
        if getattr(g, "dg", False) or getattr(g2, "dg", False) or g.degree != g2.degree or g.degree < 1 or g.degree > 3:
            return None
        if any(not np.array_equal(a, b) for a, b in zip(g.axes, g2.axes)):
            return None
        P = int(g.degree)
        nels, wlc, wlr, pr, pc, keys = [], [], [], [], [], []
        for k in range(3):
            nel = len(g.vertices[k]) - 1
            psr, psc = int(kx_row.basis.splines[k].p), int(kx_col.basis.splines[k].p)
            wr, wc = local_weights_padded(kx_row.M1[k], P, nel, psr), local_weights_padded(kx_col.M1[k], P, nel, psc)
            if wr is None or wc is None:
                return None
            nels.append(nel)
            wlr.append(wr)
            wlc.append(wc)
            pr.append(psr)
            pc.append(psc)
            keys.append((P, nel, psr, psc, _digest(kx_row.M1[k]), _digest(kx_col.M1[k])))
        return P, nels, wlc, (wlr, pr, pc), keys

    @staticmethod
    def for_pair(kx_row, kx_col):
        """plan of block (row basis, column basis); the square plan when both are the same object"""
        if kx_row is kx_col:
            plan = TensorPtAP.for_extraction(kx_row)
            if plan is not None:
                return plan
        if os.environ.get("TIGAR_PTAP_TENSOR", "1") == "0":
            return None
        cache = kx_row.__dict__.setdefault("_tensor_pair_plans", {})
        if id(kx_col) not in cache:
            st = TensorPtAP.structure_pair(kx_row, kx_col)
            cache[id(kx_col)] = (kx_col, _cached_plan("3d-pair", st[0], st[1], 1, st[4],
                                                      lambda: TensorPtAP(st[0], st[1], st[2], pair=st[3]))
                                 if st is not None else None)
        return cache[id(kx_col)][1]

    def k_nnz(self, ka, kb):
        """entries of the rows of K of the dof planes [ka, kb) (clipped band, Kronecker product)"""
        def widths(nel, pr, pc):
            nr, nc = nel + pr, nel + pc
            i = np.arange(nr)
            return np.minimum(nc - 1, i + pc) - np.maximum(0, i - pr) + 1
        w0, w1, w2 = [widths(n, self.pr[k], self.pc[k]) for k, n in enumerate(self.nels)]
        return int(w0.sum()) * int(w1.sum()) * int(w2[ka:kb].sum())

    def planes(self, A, a_row0, z0, z1):
        """x and y passes over the FE planes [z0, z1) of A (DeviceCSR holding whole planes from FE row a_row0
        on, global columns).  None: A does not have the element-coupling pattern (verified on the device)."""
        h = handle()
        rc = _lib.lib().tg_tensor_planes(self._h, A._h, int(a_row0), int(z0), int(z1), C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_tensor_planes")
        return TensorPlanes(h, int(z0), int(z1))

    def pack_kron_factors(self, factors):
        """ctypes arguments of ``tg_tensor_planes_kron`` for a Kronecker-sum form (``factors[t][k]``: scipy CSR 1-D
        matrices, all terms on one pattern per direction), or None when the form has more than three terms / terms on
        different patterns"""
        import scipy.sparse as sp
        from ._lib import tg_kron_dir_t
        nterms = len(factors)
        if nterms < 1 or nterms > 3 or any(len(f) != 3 for f in factors):
            return None
        arr = (tg_kron_dir_t * 3)()
        keep = []
        for k in range(3):
            pat = sp.csr_matrix(factors[0][k])
            pat.sort_indices()
            vals = []
            for t in range(nterms):
                F = sp.csr_matrix(factors[t][k])
                F.sort_indices()
                if F.shape != pat.shape or F.nnz != pat.nnz or not np.array_equal(F.indptr, pat.indptr) \
                        or not np.array_equal(F.indices, pat.indices):
                    return None
                vals.append(np.ascontiguousarray(F.data, dtype=np.float64))
            # the 1-D element-coupling pattern of the CG degree-p grid: row a couples to [lo(a), lo(a) + n(a))
            p, nfe = self.p, self.p * self.nels[k] + 1
            if pat.shape != (nfe, nfe):
                return None
            a = np.arange(nfe)
            vertex = (a % p == 0) & (a > 0)
            n = np.where(vertex & (a < nfe - 1), 2 * p + 1, p + 1)
            lo = np.where(vertex, a - p, (a // p) * p)
            want_ptr = np.concatenate([[0], np.cumsum(n)])
            if not np.array_equal(pat.indptr, want_ptr) or \
                    not np.array_equal(pat.indices, np.repeat(lo, n) + (np.arange(want_ptr[-1]) - np.repeat(want_ptr[:-1], n))):
                return None
            rp = np.ascontiguousarray(pat.indptr, dtype=np.int32)
            cl = np.ascontiguousarray(pat.indices, dtype=np.int32)
            vl = np.ascontiguousarray(np.concatenate(vals), dtype=np.float64)
            keep += [rp, cl, vl]
            arr[k].n = pat.shape[0]
            arr[k].rowptr = rp.ctypes.data_as(c_i32p)
            arr[k].col = cl.ctypes.data_as(c_i32p)
            arr[k].val = vl.ctypes.data_as(c_f64p)
        return nterms, arr, keep

    def planes_kron(self, packed, z0, z1):
        """x and y passes over the FE planes [z0, z1) of the matrix sum_t kron(F[t][2], F[t][1], F[t][0]) WITHOUT writing
        it (``packed`` from ``pack_kron_factors``): bit for bit the planes ``planes`` computes from the materialised
        matrix.  None: the 1-D patterns are not the element-coupling patterns of this patch."""
        nterms, arr, _keep = packed
        h = handle()
        rc = _lib.lib().tg_tensor_planes_kron(self._h, int(nterms), arr, int(z0), int(z1), C.byref(h))
        if rc == 100:
            return None
        check(rc, "tg_tensor_planes_kron")
        return TensorPlanes(h, int(z0), int(z1))

    def split(self, A):
        """(on_pattern, remainder) of a whole FE matrix on this node grid (tg_tensor_split): the entries of A that lie on
        the element-coupling pattern at their places in a matrix that has exactly that pattern, and the others as a CSR
        matrix.  None when A is not a square matrix on the grid."""
        on, off = handle(), handle()
        rc = _lib.lib().tg_tensor_split(self._h, A._h, C.byref(on), C.byref(off))
        if rc == 100:
            return None
        check(rc, "tg_tensor_split")
        return _dev.DeviceCSR(on), _dev.DeviceCSR(off)

    def zstage(self, pieces, ka, kb, zero_dofs=None, diag=1.0, append_to=None):
        """rows of K for the dof planes [ka, kb): a new DeviceCSR, or True when appended to the builder"""
        arr = (handle * len(pieces))(*[pc._h for pc in pieces])
        zd = np.ascontiguousarray(zero_dofs, dtype=np.int32) if zero_dofs is not None and len(zero_dofs) else None
        out = handle()
        check(_lib.lib().tg_tensor_zstage(self._h, len(pieces), arr, int(ka), int(kb),
                                          zd.ctypes.data_as(c_i32p) if zd is not None else None,
                                          zd.size if zd is not None else 0, float(diag),
                                          append_to._h if append_to is not None else None, C.byref(out)),
              "tg_tensor_zstage")
        return True if append_to is not None else _dev.DeviceCSR(out)


class TensorPtAP2D(object):
    """Plan of the tensor-pattern PtAP for a patch with TWO parametric directions and ``nfields`` fields on one scalar
    basis (csrc/tg_ptap_tensor.hip: tg_tensor2_*): the whole product in two line-walk passes, degrees 1..4."""

    def __init__(self, p, nels, wls, nfields=1, pair=None):
        """``pair``: (row-side weights, row degrees, column degrees) per direction for ONE block with different spline bases on
        its two sides (``wls`` then being the column side's, padded; ``nfields`` = 1) -- ``tg_tensor2_plan_create_pair``"""
        self.p, self.nels, self.nfields = int(p), [int(n) for n in nels], int(nfields)
        self._keep = [np.ascontiguousarray(w, dtype=np.float64) for w in wls]
        self._h = handle()
        if pair is not None:
            wlr, pr, pc = pair
            self._keep_r = [np.ascontiguousarray(w, dtype=np.float64) for w in wlr]
            arr = (tg_tensor_pair_dir_t * 2)()
            for k in range(2):
                arr[k].p = self.p
                arr[k].nel = self.nels[k]
                arr[k].pr, arr[k].pc = int(pr[k]), int(pc[k])
                arr[k].wlr = self._keep_r[k].ctypes.data_as(c_f64p)
                arr[k].wlc = self._keep[k].ctypes.data_as(c_f64p)
            check(_lib.lib().tg_tensor2_plan_create_pair(arr, C.byref(self._h)), "tg_tensor2_plan_create_pair")
            return
        arr = (tg_tensor_dir_t * 2)()
        for k in range(2):
            arr[k].p = self.p
            arr[k].nel = self.nels[k]
            arr[k].wl = self._keep[k].ctypes.data_as(c_f64p)
        check(_lib.lib().tg_tensor2_plan_create(self.nfields, arr, C.byref(self._h)), "tg_tensor2_plan_create")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().tg_tensor_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def structure(kx):
        """(p, nels, wls) if the 2-D patch of the ``KronExtraction`` has the structure of the fast path, else None"""
        if kx.d != 2:
            return None
        grid = kx.grid
        ps = [s.p for s in kx.basis.splines]
        p = ps[0]
        if any(q != p for q in ps) or p < 1 or p > 4 or getattr(grid, "dg", False) or grid.degree != p:
            return None
        nels, wls, keys = [], [], []
        for k in range(2):
            nel = len(grid.vertices[k]) - 1
            wl, key = checked_weights(kx.M1[k], p, nel)
            if wl is None:
                return None
            nels.append(nel)
            wls.append(wl)
            keys.append(key)
        kx._tensor_keys = keys
        return p, nels, wls

    @staticmethod
    def for_extraction(kx, nfields=1):
        """plan cached on the ``KronExtraction`` per number of fields (None if the patch does not qualify or
        TIGAR_PTAP_TENSOR=0)"""
        if os.environ.get("TIGAR_PTAP_TENSOR", "1") == "0":
            return None
        cache = kx.__dict__.setdefault("_tensor2_plans", {})
        if nfields not in cache:
            st = TensorPtAP2D.structure(kx)
            ok = st is not None and (2 * st[0] + 1) * nfields <= 64
            cache[nfields] = _cached_plan("2d", st[0], st[1], nfields, kx._tensor_keys,
                                          lambda: TensorPtAP2D(*st, nfields=nfields)) if ok else None
        return cache[nfields]

    @staticmethod
    def structure_pair(kx_row, kx_col):
        """(P, nels, column weights, (row weights, row degrees, column degrees), keys) when block (row basis, column basis) of a
        space on ONE 2-D Q_P node grid has the structure of the walks, else None"""
        if kx_row.d != 2 or kx_col.d != 2:
            return None
        g, g2 = kx_row.grid, kx_col.grid
        if getattr(g, "dg", False) or getattr(g2, "dg", False) or g.degree != g2.degree or g.degree < 1 or g.degree > 4:
            return None
        if any(not np.array_equal(a, b) for a, b in zip(g.axes, g2.axes)):
            return None
        P = int(g.degree)
        nels, wlc, wlr, pr, pc, keys = [], [], [], [], [], []
        for k in range(2):
            nel = len(g.vertices[k]) - 1
            psr, psc = int(kx_row.basis.splines[k].p), int(kx_col.basis.splines[k].p)
            wr, wc = local_weights_padded(kx_row.M1[k], P, nel, psr), local_weights_padded(kx_col.M1[k], P, nel, psc)
            if wr is None or wc is None:
                return None
            nels.append(nel)
            wlr.append(wr)
            wlc.append(wc)
            pr.append(psr)
            pc.append(psc)
            keys.append((P, nel, psr, psc, _digest(kx_row.M1[k]), _digest(kx_col.M1[k])))
        return P, nels, wlc, (wlr, pr, pc), keys

    @staticmethod
    def for_pair(kx_row, kx_col):
        """plan of block (row basis, column basis) of a 2-D space; the square plan when both are the same object"""
        if os.environ.get("TIGAR_PTAP_TENSOR", "1") == "0":
            return None
        if kx_row is kx_col:
            plan = TensorPtAP2D.for_extraction(kx_row, 1)
            if plan is not None:
                return plan
        cache = kx_row.__dict__.setdefault("_tensor2_pair_plans", {})
        if id(kx_col) not in cache:
            st = TensorPtAP2D.structure_pair(kx_row, kx_col)
            cache[id(kx_col)] = (kx_col, _cached_plan("2d-pair", st[0], st[1], 1, st[4],
                                                      lambda: TensorPtAP2D(st[0], st[1], st[2], 1, pair=st[3]))
                                 if st is not None else None)
        return cache[id(kx_col)][1]

    def ptap(self, A, zero_dofs=None, diag=1.0):
        """K = M^T A M with MatZeroRowsColumns fused, or None when A does not carry the element-coupling pattern in all
        of its nfields^2 blocks (verified on the device)."""
        zd = np.ascontiguousarray(zero_dofs, dtype=np.int32) if zero_dofs is not None and len(zero_dofs) else None
        out = handle()
        rc = _lib.lib().tg_tensor2_ptap(self._h, A._h, zd.ctypes.data_as(c_i32p) if zd is not None else None,
                                        zd.size if zd is not None else 0, float(diag), C.byref(out))
        if rc == 100:
            return None
        check(rc, "tg_tensor2_ptap")
        return _dev.DeviceCSR(out)
