"""
Sum-factorised M^T A M for extraction operators that are Kronecker products.

For a tensor-product B-spline the extraction matrix produced by ``generateM`` is
M = M_z (x) M_y (x) M_x (direction 0 fastest) whenever the ``abs(v) > eps`` filter of
tIGAr/common.py:1569 removed nothing but exact zeros -- which is checked, not assumed.  Then

    M = P_x P_y P_z,   P_x = I (x) I (x) M_x,  P_y = I (x) M_y (x) I,  P_z = M_z (x) I (x) I

and K = M^T A M = P_z^T ( P_y^T ( P_x^T A P_x ) P_y ) P_z : three general PtAP calls (the same
``k_ptap`` kernel, arbitrary FE matrix A) whose operators have <= p+1 entries per row instead
of (p+1)^d.  Hash-accumulate work drops from sum_i |supp_i| * nnz(A_r) * ... to roughly
(p+1) * nnz per stage -- 5.4x fewer LDS accumulations at 3-D p=3, 1.8x at p=2.  A is never assumed
to have any structure.  Values agree with the one-shot product to rounding (entries of M are
(Nu*Nv)*Nw either way); patterns are identical.
"""
import os

import numpy as np
import scipy.sparse as sp

from . import device as _dev


class KronExtraction(object):
    """1-D extraction matrices M_k (FE nodes x spline functions) of a tensor BSpline on its node
    grid, taken from the device evaluation tables, plus the directional operators."""

    def __init__(self, basis, grid):
        self.basis, self.grid = basis, grid
        self.d = basis.nvar
        self.M1 = []
        for k in range(self.d):
            s = basis.splines[k]
            _, idx, val = s.evalBatch(grid.axes[k])
            self.__dict__.setdefault("_tables", []).append((None, idx, val))
            n = len(grid.axes[k])
            rows = np.repeat(np.arange(n), s.p + 1)
            nz = val.ravel() != 0.0                      # exact zeros cannot pass the product filter
            M1 = sp.coo_matrix((val.ravel()[nz], (rows[nz], idx.ravel()[nz])), shape=(n, s.getNcp())).tocsr()
            M1.sort_indices()
            self.M1.append(M1)
        self.M1T = [m.T.tocsr() for m in self.M1]
        for m in self.M1T:
            m.sort_indices()
        self.nfe = [m.shape[0] for m in self.M1]
        self.ncp = [m.shape[1] for m in self.M1]
        self.nnz_product = int(np.prod([m.nnz for m in self.M1], dtype=np.float64))

    def repeated_knots(self):
        """some direction has interior knots of multiplicity > 1 (``uniformKnots(..., continuityDrop > 0)``, tIGAr/BSplines.py:
        14-38, or a knot vector written out that way): elements that bring more functions than they have nodes of their own"""
        if "_repeated" not in self.__dict__:
            self._repeated = any(len(s.multiplicities) > 2 and max(int(m) for m in s.multiplicities[1:-1]) > 1
                                 for s in self.basis.splines)
        return self._repeated

    def pattern_nnz(self, za=0, zb=None):
        """entries of the FE planes [za, zb) of a matrix on the element-coupling pattern of the node grid (what dolfin assembles
        for any form on the continuous Q_p space): product of the 1-D row-length sums; None when the grid is not such a space"""
        g = self.grid
        p = int(g.degree)
        if getattr(g, "dg", False) or p < 1:
            return None
        tot = 1
        for k in range(self.d):
            n = self.nfe[k]
            nel = (n - 1) // p
            if nel * p + 1 != n:
                return None
            a = np.arange(n)
            rn = np.where((a % p == 0) & (a > 0) & (a < n - 1), 2 * p + 1, p + 1)
            if k == self.d - 1:
                rn = rn[za:(n if zb is None else zb)]
            tot *= int(rn.sum())
        return tot

    def box_kernels_safe(self, A, za=0, zb=None):
        """A switch from the hunt for a defect of the box kernel (rows of K short by their last entries when a coupling
        added by hand met a direction with repeated knots; root cause and fix: csrc/tg_ptap_box.hip, the flags of the
        second buffer).  ``TIGAR_BOX_GUARD=1`` restores the detour of round 6's first fix: with repeated knots only a matrix
        that has exactly the pattern's number of entries takes the box / line kernels."""
        if os.environ.get("TIGAR_BOX_GUARD", "0") != "1" or not self.repeated_knots():
            return True
        want = self.pattern_nnz(za, zb)
        return want is not None and not A.is_loose() and A.nnz == want

    def products_stay_above(self, eps):
        """Sufficient condition for M == kron(M_k) entrywise WITHOUT building M: every product of stored 1-D
        entries stays above the filter threshold of generateM (abs(v) > eps, tIGAr/common.py:1569)."""
        mins = [float(np.min(np.abs(m.data))) if m.nnz else 0.0 for m in self.M1]
        return bool(np.prod(mins) > eps)

    def columns_ascending(self):
        """no periodic wrap: the functions of every 1-D node appear in ascending index order (then candidate order
        = column order and the Kronecker rows come out sorted)"""
        for k in range(self.d):
            _, idx, _ = self._tables[k]
            if np.any(np.diff(idx, axis=1) <= 0):
                return False
        return True

    def columns_distinct(self):
        """the functions of every 1-D node are pairwise different (a periodic direction with fewer functions than p+1
        would name one twice: PETSc's INSERT keeps the last value there, a Kronecker factor cannot).  With distinct
        functions the 1-D rows -- stored sorted by function index -- reproduce the reference's rows also under
        periodic wrap: the filtered, column-sorted tensor row IS the Kronecker product of the sorted 1-D rows."""
        for k in range(self.d):
            _, idx, _ = self._tables[k]
            if np.any(np.diff(np.sort(idx, axis=1), axis=1) == 0):
                return False
        return True

    # ---- periodic directions: the space before the identification of the wrapped functions --------------------
    def unwrapped(self):
        """The extraction onto the UNWRAPPED space of a patch with periodic directions, or None when no direction wraps
        (or one cannot be unwrapped).

        A periodic B-spline direction (tIGAr/BSplines.py:204-212, 246-260: ghost knots continue the knot vector, ncp =
        len(knots) - multiplicity of the first) numbers the p+1 functions of a span ``(i - p + q) mod ncp``.  Before the
        ``mod`` the functions of consecutive spans form the same chain e, e+1, ... as on an open knot vector with simple
        interior knots: nel + p functions, the last p of which are the first p again.  So M = M_u R with M_u the
        extraction onto that chain -- which has exactly the structure of the tensor line walks (``tensorptap.py``) -- and
        R the 0/1 identification matrix, one entry per row:  K = M^T A M = R^T (M_u^T A M_u) R.  ``fold`` applies R.

        The unwrapped function index of every node is found from the node tables alone (monotone continuation of the
        first function along the nodes), not from a convention about where the numbering starts."""
        if "_unwrapped" in self.__dict__:
            return self._unwrapped
        self._unwrapped = None
        if self.columns_ascending() or not self.columns_distinct():
            return None
        ku = KronExtraction.__new__(KronExtraction)
        ku.basis, ku.grid, ku.d = self.basis, self.grid, self.d
        ku.M1, ku._tables, maps = [], [], []
        for k in range(self.d):
            _, idx, val = self._tables[k]
            idx = np.asarray(idx, dtype=np.int64)
            val = np.asarray(val, dtype=np.float64)
            ncp, p = self.ncp[k], idx.shape[1] - 1
            if np.all(np.diff(idx, axis=1) > 0):
                ku.M1.append(self.M1[k])
                ku._tables.append(self._tables[k])
                maps.append(np.arange(ncp, dtype=np.int64))
                continue
            if np.any((idx[:, 1:] - idx[:, :-1]) % ncp != 1):
                return None                               # not a chain of consecutive functions
            step = np.concatenate([[0], (idx[1:, 0] - idx[:-1, 0]) % ncp])
            if np.any(step > p + 1):
                return None
            g = np.cumsum(step)[:, None] + np.arange(p + 1, dtype=np.int64)[None, :]
            n, n_u = idx.shape[0], int(g.max()) + 1
            rows = np.repeat(np.arange(n), p + 1)
            nz = val.ravel() != 0.0
            M1u = sp.coo_matrix((val.ravel()[nz], (rows[nz], g.ravel()[nz])), shape=(n, n_u)).tocsr()
            M1u.sort_indices()
            fold = (np.arange(n_u, dtype=np.int64) + int(idx[0, 0])) % ncp
            R1 = sp.csr_matrix((np.ones(n_u), fold, np.arange(n_u + 1)), shape=(n_u, ncp))
            if abs(M1u @ R1 - self.M1[k]).max() != 0.0:
                return None
            ku.M1.append(M1u)
            ku._tables.append((None, g, val))
            maps.append(fold)
        ku.M1T = [m.T.tocsr() for m in ku.M1]
        for m in ku.M1T:
            m.sort_indices()
        ku.nfe = [m.shape[0] for m in ku.M1]
        ku.ncp = [m.shape[1] for m in ku.M1]
        ku.nnz_product = int(np.prod([m.nnz for m in ku.M1], dtype=np.float64))
        ku.fold_maps, ku.wrapped_ncp = maps, list(self.ncp)
        ku._unwrapped = None
        self._unwrapped = ku
        return ku

    def fold_operators(self, nfields=1):
        """(R, R^T) on the device for an UNWRAPPED extraction (``unwrapped``): R[g, fold(g)] = 1, unwrapped dofs x dofs,
        direction 0 fastest, ``nfields`` fields numbered field after field; kept on the object"""
        cache = self.__dict__.setdefault("_fold_ops", {})
        if nfields not in cache:
            m = self.fold_maps[0]
            stride = self.wrapped_ncp[0]
            for k in range(1, self.d):
                m = (m[None, :] + stride * self.fold_maps[k][:, None]).ravel()
                stride *= self.wrapped_ncp[k]
            n_u, n = m.size, stride
            if nfields > 1:
                m = (m[None, :] + n * np.arange(nfields, dtype=np.int64)[:, None]).ravel()
            R = sp.csr_matrix((np.ones(m.size), m, np.arange(m.size + 1)), shape=(n_u * nfields, n * nfields))
            RT = R.T.tocsr()
            RT.sort_indices()
            cache[nfields] = (_dev.DeviceCSR.from_scipy(R), RT, {})
        return cache[nfields]

    def fold(self, K_u, zero_dofs=None, diag=1.0, planes=None, nfields=1):
        """K = R^T K_u R with MatZeroRowsColumns applied (tIGAr/common.py:1196-1204): the rows of the wrapped functions
        are added to the rows they are identified with, the columns renamed -- one pass of the general PtAP kernels
        over K_u (its cost: one accumulation per entry).  ``planes`` = (k0, k1): K_u holds the rows of the dof planes
        [k0, k1) of the last direction only (global unwrapped columns), and so does the result (a direction that wraps
        cannot be the slab direction)."""
        R, RT_host, rt_cache = self.fold_operators(nfields)
        if planes is None:
            a_row0 = mt_row0 = 0
            key = None
        else:
            assert nfields == 1 and np.array_equal(self.fold_maps[-1], np.arange(self.ncp[-1]))
            pl_u = int(np.prod(self.ncp[:-1])) if self.d > 1 else 1
            pl = int(np.prod(self.wrapped_ncp[:-1])) if self.d > 1 else 1
            a_row0, mt_row0 = planes[0] * pl_u, planes[0] * pl
            key = (int(planes[0]), int(planes[1]))
        if key not in rt_cache:
            if len(rt_cache) > 4:
                rt_cache.clear()
            rt_cache[key] = _dev.DeviceCSR.from_scipy(RT_host if key is None else RT_host[key[0] * pl:key[1] * pl])
        RT = rt_cache[key]
        if K_u.is_loose():
            K_u = K_u.compact()
        # The first product on a pattern goes through the general kernels with R as the operator (96^3 p=3 periodic in all
        # directions: 36 ms of a 45 ms product -- a row of K gathers ONE row of K_u almost everywhere, so every key is new to
        # the row's table); it gives the pattern of K, and a plan then keeps the PLACE of every entry of K_u in its row of K:
        # all later products (K_u of the tensor walks has the same closed-form pattern every time) are one pass over K_u
        # without look-up (tg_foldplan_*).  The values of the first product come from the plan as well: same order of additions.
        plans = self.__dict__.setdefault("_fold_plans", {})
        pkey = (key, nfields, K_u.shape, K_u.nnz)
        fp = plans.get(pkey)
        if fp is not None:
            K = fp.apply(K_u, zero_dofs, diag)
            if K is not None:
                return K
        plan = _dev.ptap_symbolic(K_u, R, RT, a_row0, 0, mt_row0)
        K = _dev.ptap_numeric(plan, K_u, R, RT, zero_dofs, diag)
        if os.environ.get("TIGAR_FOLD_PLAN", "1") != "0" and K_u.nnz <= 1.5e9:
            if len(plans) > 4:
                plans.clear()
            fp = _dev.FoldPlan.create(K_u, R, RT, a_row0, mt_row0, K)
            plans[pkey] = fp
            if fp is not None:
                K2 = fp.apply(K_u, zero_dofs, diag)
                if K2 is not None:
                    return K2
        return K

    def is_exact_for(self, M_nnz, eps):
        """True if generateM's filter dropped only exact zeros, i.e. M == kron(M_k) entrywise."""
        small = any(np.any(np.abs(m.data) <= eps) for m in self.M1)
        return (not small) and int(M_nnz) == self.nnz_product

    # ---- index spaces: directions in `done` are spline-sized, the others FE-sized ---------------
    def dims(self, done):
        return [self.ncp[k] if k in done else self.nfe[k] for k in range(self.d)]

    def plane(self, done):
        """entries per plane of the last direction in the space where `done` is contracted"""
        dm = self.dims(done)
        return int(np.prod(dm[:-1])) if self.d > 1 else 1

    def _factors(self, done, group, transpose):
        """Operator contracting the directions in `group` (M_k there, identities elsewhere) acting
        on the space where `done` is already contracted."""
        dm = self.dims(done)
        fac = []
        for k in range(self.d):
            if k in group:
                fac.append(self.M1[k].T.tocsr() if transpose else self.M1[k])
            else:
                fac.append(sp.identity(dm[k], format="csr"))
        return fac

    def P(self, done, group, row0=None, row1=None):
        return _dev.kron_csr_rect(self._factors(done, group, False), row0, row1)

    def PT(self, done, group, row0=None, row1=None):
        return _dev.kron_csr_rect(self._factors(done, group, True), row0, row1)


def default_groups(d, p):
    """Which directions to contract together.  One-shot ([[0..d-1]]) is the plain PtAP with the
    workgroup-per-row box kernel; one direction at a time runs the wave-per-run line kernel
    (k_ptap_line) three times with 2.5x fewer products at 3-D p=3.  Measured on MI355X: the one-shot
    product wins in 2-D (3.1 vs 3.2 ms at 256^2 p=4), x|y|z wins in 3-D (ptap 1.60 s vs 2.06 s per step
    at 256^3 p=3 when the line kernel was new; at 128^3 p=2 the one-shot product won then, 51.6 vs 57.6 ms,
    and loses now: 51.5 vs 35.0 ms)."""
    import os
    env = os.environ.get("TIGAR_PTAP_GROUPS")            # e.g. "0;1;2" or "0,1;2" (experiments)
    if env:
        groups = [[int(c) for c in g.split(",") if int(c) < d] for g in env.split(";")]
        groups = [g for g in groups if g]
        if sorted(sum(groups, [])) == list(range(d)) and (d - 1) in groups[-1]:
            return groups
    if d == 3 and p >= 2:
        return [[0], [1], [2]]
    return [list(range(d))]


def contract(kx, cur, done, group, a_planes, c_planes, out_rows, zero_dofs=None, diag=1.0, intermediate=False,
             append_to=None, box=True):
    """One contraction stage: rows ``out_rows`` (global row range in the space after the stage)
    of  P^T cur P,  where ``cur`` holds the planes ``a_planes`` of the current space (global
    columns within the planes ``c_planes``) and P contracts the directions in ``group``."""
    pl_in = kx.plane(done)
    import os
    # the box / line kernels address an output row's operands as ONE interval per direction: a patch with a periodic
    # direction (supports that wrap around) takes the general kernels in every stage
    if "_wraps" not in kx.__dict__:
        kx._wraps = not kx.columns_ascending()
    if box and os.environ.get("TIGAR_PTAP_BOX", "1") != "0" and not kx._wraps:
        dims_in = kx.dims(done)
        factors = [kx.M1[k] if k in group else None for k in range(kx.d)]
        out = _dev.ptap_kron(cur, a_planes[0] * pl_in, dims_in, factors, out_rows[0], out_rows[1], zero_dofs, diag,
                             intermediate=intermediate, append_to=append_to)
        if out is not None:
            return out               # (True: the rows went straight into the builder)
    if cur.is_loose():
        cur = cur.compact()              # the general kernel needs canonical rows
    MT = kx.PT(done, group, out_rows[0], out_rows[1])
    Pm = kx.P(done, group, c_planes[0] * pl_in, c_planes[1] * pl_in)
    try:
        plan = _dev.ptap_symbolic(cur, Pm, MT, a_planes[0] * pl_in, c_planes[0] * pl_in, out_rows[0])
        return _dev.ptap_numeric(plan, cur, Pm, MT, zero_dofs, diag)
    except _dev.TigarHipError as e:
        nlast = kx.dims(done)[-1]
        if "does not cover" not in str(e) or tuple(c_planes) == (0, nlast):
            raise
    # a row block of a streamed / distributed product whose columns reach beyond the planes its elements couple to (a
    # coupling added by hand, tIGAr/common.py:1194-1195 takes any A; the symbolic pass samples rows, so the numeric
    # product may be the one that meets it): the operator on all planes
    del Pm
    Pm = kx.P(done, group, 0, nlast * pl_in)
    plan = _dev.ptap_symbolic(cur, Pm, MT, a_planes[0] * pl_in, 0, out_rows[0])
    return _dev.ptap_numeric(plan, cur, Pm, MT, zero_dofs, diag)


def ptap_factored(kx, A, a_planes, c_planes, k_planes, zero_dofs=None, diag=1.0, groups=None, _split=True, stored=None):
    """K rows of the dof planes ``k_planes`` from the FE rows ``A`` (DeviceCSR holding the FE
    planes ``a_planes`` = [za,zb) of the last direction, global columns reaching the planes
    ``c_planes`` = [ca,cb)), by contracting the direction groups one after the other.  Planes
    refer to the LAST direction, which must be in the last group; for a resident single block pass
    the full ranges."""
    d = kx.d
    za, zb = a_planes
    k0, k1 = k_planes
    # tensor-pattern fast path (csrc/tg_tensor_body.h): line walks without column decode; declines (None)
    # when the patch or A's pattern does not qualify, and the general stages below take over
    from .tensorptap import TensorPtAP
    plan = TensorPtAP.for_extraction(kx)
    if plan is None and not A.is_loose() and (za, zb) == (0, kx.nfe[-1]) and (k0, k1) == (0, kx.ncp[-1]) \
            and os.environ.get("TIGAR_PTAP_UNWRAP", "1") != "0":
        # periodic directions: the line walks on the unwrapped space, then the identification of the wrapped functions
        ku = kx.unwrapped()
        if ku is not None and TensorPtAP.for_extraction(ku) is not None:
            K_u = ptap_factored(ku, A, a_planes, c_planes, (0, ku.ncp[-1]), None, 1.0, None, _split)
            return ku.fold(K_u, zero_dofs, diag)
    if plan is not None and not A.is_loose():
        piece = plan.planes(A, za * kx.plane(set()), za, zb)
        if piece is not None:
            return plan.zstage([piece], k0, k1, zero_dofs, diag)
        # Another pattern (couplings added by hand -- contact, constraints -- or entries missing).  A whole resident
        # matrix is split: what lies on the pattern goes through the line walks, the few entries outside through the
        # general stages below, and the two products are added on the union of their patterns (the structural pattern
        # of the whole product); the boundary conditions follow on the sum.  Worth it while the remainder is small.
        whole = (za, zb) == (0, kx.nfe[-1]) and (k0, k1) == (0, kx.ncp[-1]) and A.shape[0] == A.shape[1]
        if _split and whole and os.environ.get("TIGAR_PTAP_SPLIT", "1") != "0":
            parts = plan.split(A)
            # (`on` comes back on the FULL element-coupling pattern: where A has no entry there, a zero is stored.  An
            #  assembled FE matrix has them all; one that lacks some would get stored zeros in K that MatPtAP's symbolic
            #  product does not create -- such a matrix takes the general stages)
            if parts is not None and parts[1].nnz <= 0.25 * A.nnz and A.nnz - parts[1].nnz == parts[0].nnz:
                on, off = parts
                K = plan.zstage([plan.planes(on, 0, za, zb)], k0, k1)
                if off.nnz:
                    if stored is not None:
                        # few entries anywhere in the matrix: the hash kernels (any sparsity) with the stored M, M^T
                        M, MT = stored
                        Kr = _dev.ptap_numeric(_dev.ptap_symbolic(off, M, MT), off, M, MT)
                    else:
                        Kr = ptap_factored(kx, off, a_planes, c_planes, k_planes, None, diag, groups, _split=False)
                    K = K.add(Kr)
                if zero_dofs is not None and len(zero_dofs):
                    K.zero_rows_cols(np.asarray(zero_dofs, dtype=np.int32), diag)
                return K
    if groups is None:
        groups = [[k] for k in range(d)]
    assert sorted(sum(groups, [])) == list(range(d)) and (d - 1) in groups[-1]
    cur = A
    done = set()
    box = kx.box_kernels_safe(A, za, zb)
    for gi, group in enumerate(groups):
        last = (gi == len(groups) - 1)
        after = done | set(group)
        pl_out = kx.plane(after)
        out_rows = (k0 * pl_out, k1 * pl_out) if last else (za * pl_out, zb * pl_out)
        cur = contract(kx, cur, done, group, a_planes, c_planes, out_rows, zero_dofs if last else None, diag,
                       intermediate=not last, box=box)
        done = after
    return cur
