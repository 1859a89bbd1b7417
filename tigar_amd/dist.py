"""
z-slab decomposition of the extraction hot path -- used both for
multi-GPU runs (one process per GPU, slabs distributed over ranks, RCCL halo exchange +
scalar all-reduces in the Krylov solve) and for streaming a problem whose M and A do not fit
in HBM through one GPU slab by slab (only K is kept resident).

The reference gets its row-block distribution from dolfin's mesh partition and PETSc's
parallel MatPtAP / VecScatter (tIGAr/common.py:1194-1195, 1255-1261 [ext]).  Here IGA dof
index = i + j*ncp_x + k*ncp_x*ncp_y (tIGAr/BSplines.py:357-358), so contiguous dof ranges
are slabs of "planes" along the last parametric direction; everything a slab of K rows needs
(rows of M^T, rows of A, rows of M) is local in that direction, and its extent follows from
the 1-D knot vector alone.  ``ZSlabLayout`` is pure host arithmetic (CPU-testable).
"""
import os
import numpy as np


def split_range(n, parts):
    """Balanced contiguous split of range(n): list of (start, stop), len == parts."""
    base, extra = divmod(n, parts)
    out, s = [], 0
    for r in range(parts):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


class ZSlabLayout(object):
    """Index arithmetic of slabs along the LAST parametric direction of a tensor B-spline.

    knots / p: the open knot vector and degree of that direction; fe_nodes: its 1-D FE node
    coordinates (CG Lagrange degree ``q`` on the unique-knot mesh, vertex nodes on knots);
    plane_dofs / plane_fe: number of spline dofs / FE nodes per plane (product over the other
    directions)."""

    def __init__(self, knots, p, fe_nodes, q, plane_dofs, plane_fe):
        self.knots = np.asarray(knots, dtype=np.float64)
        self.p, self.q = int(p), int(q)
        self.nodes = np.asarray(fe_nodes, dtype=np.float64)
        self.ncp = len(self.knots) - (self.p + 1)
        if not (np.all(self.knots[:p + 1] == self.knots[0]) and np.all(self.knots[-(p + 1):] == self.knots[-1])):
            raise ValueError("z-slab decomposition needs an open knot vector in the slab direction")
        self.nfe = len(self.nodes)
        self.nel = (self.nfe - 1) // self.q
        if self.nel * self.q + 1 != self.nfe:
            raise ValueError("FE nodes are not a CG degree-%d grid" % self.q)
        self.plane_dofs, self.plane_fe = int(plane_dofs), int(plane_fe)
        # support of basis k = [knots[k], knots[k+p+1]]: FE planes whose node lies inside it
        lo = np.searchsorted(self.nodes, self.knots[:self.ncp], side="left")
        hi = np.searchsorted(self.nodes, self.knots[p + 1:p + 1 + self.ncp], side="right")
        self.sup_lo, self.sup_hi = lo, hi          # FE planes [lo, hi) of every dof plane
        # first basis function that is non-zero on FE plane a: left-biased span - p
        span = np.searchsorted(self.knots, self.nodes, side="left") - 1
        span = np.clip(span, p, len(self.knots) - p - 2)
        self.first_basis = span - p

    # ---- extents ------------------------------------------------------------------------
    def fe_planes_of_dofs(self, k0, k1):
        """FE planes [za, zb) holding the rows of M^T rows k0..k1 (the support of the slab)."""
        return int(self.sup_lo[k0:k1].min()), int(self.sup_hi[k0:k1].max())

    def fe_planes_coupled(self, za, zb):
        """FE planes [ca, cb) that rows za..zb of an FE matrix on the CG grid couple to
        (all nodes of every element touching the range)."""
        q = self.q
        e_lo = (za - 1) // q if (za > 0 and za % q == 0) else za // q
        zl = zb - 1
        e_hi = min(self.nel - 1, zl // q)
        return int(max(0, q * e_lo)), int(min(self.nfe, q * (e_hi + 1) + 1))

    def dof_halo(self, k0, k1):
        """(halo_lo, halo_hi) in planes: dof planes outside [k0,k1) that rows of K couple to
        (supports sharing an element with a support of the slab)."""
        U, p = self.knots, self.p
        lo_knot, hi_knot = U[k0], U[k1 - 1 + p + 1]
        ks = np.arange(self.ncp)
        overl = (U[ks] < hi_knot) & (U[ks + p + 1] > lo_knot)
        idx = np.flatnonzero(overl)
        return int(k0 - idx.min()), int(idx.max() + 1 - k1)

    def owned_fe_planes(self, k0, k1):
        """FE planes whose prolongation rows u = M U are computed by the owner of dof planes
        [k0,k1): plane a belongs to the owner of the first basis function non-zero on it, so
        its columns lie in [k0, k1 + p)."""
        a = np.flatnonzero((self.first_basis >= k0) & (self.first_basis < k1))
        if a.size == 0:
            return 0, 0
        return int(a.min()), int(a.max() + 1)

    # ---- row ranges (global indices) of everything a slab of K rows needs ------------------
    def slab(self, k0, k1):
        if k1 <= k0:                      # (a rank without planes of this basis: fields of unequal plane counts share one split)
            return {"dofs": (k0 * self.plane_dofs, k0 * self.plane_dofs), "a_rows": (0, 0), "m_rows": (0, 0), "halo": (0, 0),
                    "u_rows": (0, 0)}
        za, zb = self.fe_planes_of_dofs(k0, k1)
        ca, cb = self.fe_planes_coupled(za, zb)
        hl, hh = self.dof_halo(k0, k1)
        oa, ob = self.owned_fe_planes(k0, k1)
        return {
            "dofs": (k0 * self.plane_dofs, k1 * self.plane_dofs),           # rows of M^T / K
            "a_rows": (za * self.plane_fe, zb * self.plane_fe),              # rows of A
            "m_rows": (ca * self.plane_fe, cb * self.plane_fe),              # rows of M
            "halo": (hl * self.plane_dofs, hh * self.plane_dofs),            # Krylov vector halo (dofs)
            "u_rows": (oa * self.plane_fe, ob * self.plane_fe),              # prolongation rows
        }


def layout_for(bspline, grid):
    """ZSlabLayout of a tigar_amd BSpline scalar basis on its TensorNodeGrid."""
    s = bspline.splines[-1]
    shape = grid.shape()
    plane_fe = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    ncps = [sp.getNcp() for sp in bspline.splines]
    plane_dofs = int(np.prod(ncps[:-1])) if len(ncps) > 1 else 1
    return ZSlabLayout(s.knots, s.p, grid.axes[-1], grid.degree, plane_dofs, plane_fe)


def pick_sub_planes(d, p, nel, planes_mine, free_bytes):
    """dof planes per sub-slab so that one slab of A and the PtAP temporaries use at most about half
    of the free HBM (K needs the rest).  Measured at 256^3 p=3 with the 3/4-of-HBM allocator pool, per step:
    5 planes 1.59 s of input+PtAP, 8: 1.53 s, 12: 1.44 s, 16: 1.41 s with 75 GB still free at the end of a
    step, 20: allocation failures and pool trimming start, 24: 5.6 s.  (The sliced copy of K's values that
    lives during the Krylov solve, 47 GB at 256^3 p=3, is stored in idle blocks of the allocator's pool --
    the PtAP temporaries sized here -- so it needs no budget of its own.)  The tensor-pattern passes of round 2 need
    less per plane (A rows + two dense intermediates, 1.6 GB per FE plane at 256^3 p=3) and are insensitive to the
    choice: 12 planes (what this returns there) 1.145 s per step, 16: 1.148 s, 20: the allocator starts trimming."""
    nfe1 = nel * p + 1
    plane_fe = nfe1 ** (d - 1)
    nnzA_plane = plane_fe * ((2 * p + 1) ** d) * 0.55 * 12.0        # bytes per FE plane, generous
    nnzM_plane = plane_fe * ((p + 1) ** d) * 12.0
    per_dof_plane = p * (nnzA_plane + 2.2 * nnzM_plane) + (nel + p) ** (d - 1) * ((2 * p + 1) ** d) * 12.0 * 2
    fixed = (2 * p * p + 2) * (nnzA_plane + 2.2 * nnzM_plane)
    budget = 0.5 * free_bytes - fixed
    return int(max(1, min(planes_mine, budget // per_dof_plane)))


class _ChunkRows(object):
    """Rows [g0, g1) of a matrix that arrives as CHUNKS of rows to be ADDED (the element chunks of ``SlabHotPath``): rows below
    ``done_row`` after a chunk are final and go to the result (a ``CSRBuilder`` sized from the densest plane of the first rows),
    the rows above are carried and added to the next chunk's on the union pattern."""

    def __init__(self, dev, g0, g1, ncols, plane_rows):
        self.dev, self.g0, self.g1, self.ncols, self.pd = dev, int(g0), int(g1), int(ncols), int(plane_rows)
        self.builder, self.single = None, None
        self.pending, self.pend0 = None, 0        # rows [pend0, pend0 + pending.shape[0]): sums over the chunks so far
        self.emitted = self.g0

    def _rows(self, Kb, base, r0, r1):
        return Kb.block(r0 - base, r1 - base, 0, self.ncols)

    def _emit(self, Kb, base, r0, r1):
        """rows [r0, r1) of the block Kb (whose first row is ``base``) are final: those of this rank go to the result"""
        r0, r1 = max(r0, self.g0, self.emitted), min(r1, self.g1)
        if r1 <= r0:
            return
        assert r0 == self.emitted, "element chunks: rows must come in order"
        blk = Kb if (r0 == base and r1 == base + Kb.shape[0]) else self._rows(Kb, base, r0, r1)
        if r0 == self.g0 and r1 == self.g1:
            self.single = blk
        else:
            if self.builder is None:
                # capacity from the densest part of the first rows: their last dof plane (rows near the patch boundary are
                # shorter; a builder that has to grow allocates and copies K a second time -- 75 GB at 256^3 p = 3)
                nr, pd = blk.shape[0], self.pd
                last = (blk.nnz - blk.rowptr_at(nr - pd)) / float(pd) if nr >= pd else blk.nnz / float(max(1, nr))
                est = int(max(blk.nnz / float(max(1, nr)), last) * (self.g1 - self.g0) * 1.01) + 1024
                self.builder = self.dev.CSRBuilder(self.g1 - self.g0, self.ncols, est)
            self.builder.append(blk)
        self.emitted = r1

    def add_chunk(self, Kc, d0, d1, done_row):
        """``Kc``: rows [d0, d1) of a chunk's contribution; rows below ``done_row`` receive nothing after it"""
        if self.pending is not None:
            pend0, pending = self.pend0, self.pending
            p1 = pend0 + pending.shape[0]
            if d0 > pend0:                                   # (rows below the new chunk: nothing more comes for them)
                self._emit(pending, pend0, pend0, min(d0, p1))
            ov0, ov1 = max(d0, pend0), min(p1, d1)
            parts = []
            if ov1 > ov0:
                parts.append(self._rows(pending, pend0, ov0, ov1).add(self._rows(Kc, d0, ov0, ov1)))
            if d1 > max(p1, d0):
                parts.append(self._rows(Kc, d0, max(p1, d0), d1))
            if p1 > d1:
                parts.append(self._rows(pending, pend0, max(d1, pend0), p1))
            start = min(ov0, max(p1, d0)) if ov1 > ov0 else max(p1, d0)
            self.pending, self.pend0 = (parts[0] if len(parts) == 1 else self.dev.csr_vstack(parts)), start
        else:
            self.pending, self.pend0 = Kc, d0
        p1 = self.pend0 + self.pending.shape[0]
        cut = min(max(done_row, self.pend0), p1)
        if cut > self.pend0:
            self._emit(self.pending, self.pend0, self.pend0, cut)
            self.pending = self._rows(self.pending, self.pend0, cut, p1) if cut < p1 else None
            self.pend0 = cut

    def finish(self):
        if self.pending is not None:
            self._emit(self.pending, self.pend0, self.pend0, self.pend0 + self.pending.shape[0])
            self.pending = None
        if self.emitted != self.g1:
            raise RuntimeError("element chunks: rows %d .. %d of this rank were not produced" % (self.emitted, self.g1))
        return self.single if self.builder is None else self.builder.finish()


class _TensorDeclined(Exception):
    """the FE matrix does not have the element-coupling pattern the tensor-pattern PtAP needs"""


class SlabHotPath(object):
    """The hot path of one rank, streamed through HBM in sub-slabs of dof planes:

        for every sub-slab:  M^T rows, M rows  <- extraction kernels (row ranges)
                             A rows            <- ``a_rows(row0, row1)`` (FE-side input)
                             K rows            <- PtAP on the three row blocks
                             (M^T b) rows      <- slab SpMV
        K_loc = vstack(K rows);  Krylov solve with halo exchange;  u_loc = M_own * U

    With world == 1 and one sub-slab this is exactly the single-GPU path."""

    def __init__(self, basis, grid, rank=0, world=1, comm=None, sub_planes=None, eps=1e-15, factored=None, kx=None, planes=None,
                 resident_blocks=1):
        """``planes``: the dof planes [k0, k1) of this rank when they are not the balanced split of the basis' own planes
        (fields on different bases share one split of the plane index, ``FieldListSlabPath``); ``resident_blocks``: how many
        blocks the size of this engine's K stay on the device while it streams (nF x nF field blocks + their stacked copy):
        ``sub_planes='auto'`` sizes the sub-slabs for what is left"""
        from . import device as dev
        from .kronptap import KronExtraction
        self.dev = dev
        self.kx = kx if kx is not None else KronExtraction(basis, grid)
        # sum-factorised PtAP when M is exactly a Kronecker product (checked against the
        # closed-form nnz of M on the tensor grid); TIGAR_PTAP_FACTORED=0/1 overrides
        env = os.environ.get("TIGAR_PTAP_FACTORED")
        from .kronptap import default_groups
        self.groups = default_groups(basis.nvar, max(s1.p for s1 in basis.splines))
        if env is not None and env not in ("0", "1"):           # e.g. TIGAR_PTAP_FACTORED=0,1;2 or 0;1;2
            g_env = [[int(c) for c in g.split(",") if int(c) < basis.nvar] for g in env.split(";")]
            g_env = [g for g in g_env if g]
            if sorted(sum(g_env, [])) == list(range(basis.nvar)):
                self.groups = g_env
        if factored is True and len(self.groups) == 1:
            self.groups = [[k] for k in range(basis.nvar)]
        explicit = env is not None and env not in ("0", "1")
        self.factored = (factored if factored is not None else (env != "0" if env is not None else True))
        self.basis, self.grid = basis, grid
        # sufficient condition for M == kron(M_k) entrywise without building anything: every product of
        # 1-D entries stays above the filter threshold (generateM drops abs(v) <= eps)
        mins = [float(np.min(np.abs(m.data))) if m.nnz else 0.0 for m in self.kx.M1]
        self.kron_exact = bool(np.prod(mins) > eps) and os.environ.get("TIGAR_TENSOR_APPLY", "1") != "0"
        self.rank, self.world, self.comm = rank, world, comm
        self.eps = eps
        self.layout = layout_for(basis, grid)
        self.k0, self.k1 = split_range(self.layout.ncp, world)[rank] if planes is None else (int(planes[0]), int(planes[1]))
        self._sub_auto = sub_planes == "auto" and not os.environ.get("TIGAR_SUB_PLANES")
        if sub_planes == "auto" and os.environ.get("TIGAR_SUB_PLANES"):
            sub_planes = int(os.environ["TIGAR_SUB_PLANES"])          # (experiments)
        if sub_planes == "auto":
            # what the path can use: free device memory plus what the library's caching allocator holds idle
            free_b = dev.mem_info()[0] + dev.pool_stats()[0]
            if world > 1:
                # ranks that share a GPU (fewer devices than local ranks) all see the same free memory: each takes its share
                # (eight ranks of cfg3 on one GPU ran out of memory when every one of them sized its sub-slabs for all of it)
                try:
                    local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
                    share = max(1, -(-local // max(1, dev.device_count())))
                    free_b //= share
                    if share > 1:
                        self._sub_auto = False        # (ranks sharing a device: no larger sub-slabs for the fused forms either)
                except Exception:
                    pass
            pmax = max(s1.p for s1 in basis.splines)
            nelmax = max(s1.nel for s1 in basis.splines)
            if resident_blocks > 1:
                # (K of one block: rows of this rank x (2 p + 1)^d entries of 12 bytes; pick_sub_planes leaves half of what it
                #  is given to ONE such block already)
                k_block = (self.k1 - self.k0) * float(self.layout.plane_dofs) * float((2 * pmax + 1) ** basis.nvar) * 12.0
                free_b = max(free_b - int((resident_blocks - 1) * k_block), free_b // 8)
            sub_planes = pick_sub_planes(basis.nvar, pmax, nelmax, self.k1 - self.k0, free_b)
        self.sub_planes = sub_planes or (self.k1 - self.k0)
        self.mine = self.layout.slab(self.k0, self.k1)
        self.ncp = basis.getNcp()
        self.n_fe = grid.num_nodes()
        if comm is not None and world > 1:
            g0, g1 = self.mine["dofs"]
            comm.set_slab(g0, g1, self.mine["halo"][0], self.mine["halo"][1], self.ncp)

    def sub_slabs(self, sub_planes=None):
        """dof-plane ranges of the sub-slabs: as few as ``sub_planes`` allows, of (nearly) equal size -- a short last
        one costs its launches and host round trip for little work (32 planes of a rank in sub-slabs of 14: 14+14+4)"""
        n = self.k1 - self.k0
        if n <= 0:
            return []
        parts = -(-n // max(1, sub_planes or self.sub_planes))
        return [(self.k0 + (n * q) // parts, self.k0 + (n * (q + 1)) // parts) for q in range(parts)]

    def assemble(self, a_rows, b_rows, zero_dofs, diag=1.0, timers=None, a_factors=None, col=None):
        """K_loc (rows of this rank, global columns, BCs applied) and rhs_loc = (M^T b)_loc.
        ``a_rows(r0, r1)`` / ``b_rows(r0, r1)`` return the FE matrix rows (DeviceCSR, global
        columns) / FE vector entries (DeviceVector) of global FE rows [r0, r1).
        ``col``: the ``KronExtraction`` of the COLUMN side when it is another basis than this engine's (block (f, g) of a
        space whose fields sit on different bases over one node grid: K_fg = M_f^T A_fg M_g) -- tensor line walks only."""
        import time
        dev = self.dev
        if col is not None:
            return self._assemble_pair(a_rows, col, timers, a_factors)
        if not self.factored and os.environ.get("TIGAR_PTAP_ELEMENTS", "1") != "0":
            # nothing assumed about M or the values of A: the element split in chunks of element layers (csrc/tg_elemsplit.hip);
            # an FE matrix with an entry between nodes of no common cell is declined and takes the row-wise stages below
            out = self._assemble_by_elements(a_rows, b_rows, zero_dofs, diag, timers)
            if out is not None:
                return out
        sp1, axes = self.basis.splines, self.grid.axes
        zero_dofs = np.asarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        with_rhs = b_rows is not None
        k_blocks, rhs_parts = [], []
        ring = {"hi": 0, "pieces": []}      # plane-local stage results kept across sub-slabs
        builder = None
        nslabs = len(self.sub_slabs())
        t = timers if timers is not None else {}

        def tick(name, t0):
            dev.sync()
            t[name] = t.get(name, 0.0) + time.perf_counter() - t0

        # Optional (TIGAR_OVERLAP=1): the FE inputs of sub-slab i+1 are requested on the library's second
        # stream before the contraction of sub-slab i is enqueued on the first (only on the fully
        # factorised path, where the rows each sub-slab needs are known up front).  Measured at
        # 256^3 p=3: the kernels do run side by side (rocprof: the 6 ms fill stretches over the whole
        # 28 ms of a sub-slab's PtAP), but both are bound by instruction issue on the same CUs -- the
        # input time disappears from the host's view (0.20 -> 0.03 s) and the PtAP grows by the same
        # amount (1.03 -> 1.18 s), with or without a low-priority producer stream.  Off by default.
        subs = self.sub_slabs()
        # tensor-pattern fast path (csrc/tg_tensor_body.h): the plane-local passes give dense B2 planes that are
        # kept in the ring, the z pass writes the rows of K at closed-form positions (exact capacity known)
        from .tensorptap import TensorPtAP
        tplan = TensorPtAP.for_extraction(self.kx) if (self.factored and self.kron_exact and not getattr(
            self, "_tensor_declined", False)) else None
        # directions other than the slab direction that wrap (periodic, tIGAr/BSplines.py:204-212): the walks on the
        # unwrapped space, this rank's rows of K_u folded at the end (kronptap.KronExtraction.unwrapped / fold)
        fold = None
        if tplan is None and self.factored and self.kron_exact and not getattr(self, "_tensor_declined", False) \
                and os.environ.get("TIGAR_PTAP_UNWRAP", "1") != "0":
            ku = self.kx.unwrapped()
            if ku is not None and np.array_equal(ku.fold_maps[-1], np.arange(ku.ncp[-1])):
                tplan = TensorPtAP.for_extraction(ku)
                fold = ku if tplan is not None else None
        ring["tensor"] = tplan
        # an FE matrix that is a Kronecker sum of 1-D matrices on the element-coupling pattern is never written: the x
        # pass forms its entries (tg_tensor_planes_kron; TIGAR_PTAP_FUSED=0 materialises the row blocks as before)
        ring["kron"] = None
        if tplan is not None and a_factors is not None and os.environ.get("TIGAR_PTAP_FUSED", "1") != "0":
            ring["kron"] = tplan.pack_kron_factors(a_factors)
            # the FE matrix is never written: a sub-slab needs its two dense intermediates only, and the automatic choice (sized
            # for materialised rows of A) can double -- cfg3: 24 planes instead of 12, MtAM 0.170 -> 0.160 s (fewer launches and
            # ring hand-overs; 32: 0.158, 16: 0.160)
            if ring["kron"] is not None and getattr(self, "_sub_auto", False):
                subs = self.sub_slabs(2 * self.sub_planes)
                nslabs = len(subs)
                self.sub_planes_used = 2 * self.sub_planes
        if tplan is not None and nslabs > 1:
            if fold is None:
                builder = dev.CSRBuilder(self.mine["dofs"][1] - self.mine["dofs"][0], self.ncp, tplan.k_nnz(self.k0, self.k1))
            else:
                pl_u = int(np.prod(fold.ncp[:-1]))
                builder = dev.CSRBuilder((self.k1 - self.k0) * pl_u, pl_u * fold.ncp[-1], tplan.k_nnz(self.k0, self.k1))
        overlap = (self.factored and self.kron_exact and len(subs) > 1
                   and os.environ.get("TIGAR_OVERLAP", "0") == "1")
        sched = []
        if overlap:
            pf0, hi = self.layout.plane_fe, 0
            for (ka, kb) in subs:
                S0 = self.layout.slab(ka, kb)
                za, zb = S0["a_rows"][0] // pf0, S0["a_rows"][1] // pf0
                lo = max(za, hi)
                sched.append((lo, zb, S0["a_rows"]))
                if zb > lo:
                    hi = zb

        def produce(i):
            lo, zb, arows = sched[i]
            dev.stream_set(1)
            try:
                # the vector first: producers of small objects tend to end with a host synchronisation
                # (uploads from host arrays), which must not sit behind the long fill kernel of the matrix
                b_i = b_rows(arows[0], arows[1]) if with_rhs else None
                A_i = a_rows(lo * self.layout.plane_fe, zb * self.layout.plane_fe) if zb > lo else None
            finally:
                dev.stream_set(0)
            return A_i, b_i

        nxt = produce(0) if overlap else None
        for islab, (ka, kb) in enumerate(subs):
            S = self.layout.slab(ka, kb)
            t0 = time.perf_counter()
            use_factored = self.factored
            tensor_mtb = use_factored and self.kron_exact
            MT = None
            if not tensor_mtb and (with_rhs or not use_factored):
                MT = dev.extract_csr_tensor_t(sp1, axes, 0, self.n_fe, self.eps, S["dofs"][0], S["dofs"][1])
            if use_factored and not tensor_mtb:
                # exactness check of the Kronecker form on this slab: nnz(M^T rows) must equal the
                # product of the 1-D counts restricted to the slab's dof planes
                lz = self.kx.M1[-1].T.tocsr()
                nz_last = int(lz.indptr[kb] - lz.indptr[ka])
                expect = nz_last * int(np.prod([m.nnz for m in self.kx.M1[:-1]], dtype=np.float64))
                use_factored = (MT.nnz == expect) and not any(np.any(np.abs(m.data) <= self.eps) for m in self.kx.M1)
            M = None
            if not use_factored:
                M = dev.extract_csr_tensor(sp1, axes, 0, self.ncp, self.eps, S["m_rows"][0], S["m_rows"][1])
            tick("extract", t0)
            t0 = time.perf_counter()
            pf = self.layout.plane_fe
            if overlap:
                dev.stream_wait(0, 1)                      # the inputs of this sub-slab are complete
                A, b = nxt
                ring["new"] = (sched[islab][0], sched[islab][1])
                nxt = produce(islab + 1) if islab + 1 < len(subs) else None
            elif use_factored:
                za, zb = S["a_rows"][0] // pf, S["a_rows"][1] // pf
                new_lo = max(za, ring["hi"])               # FE planes not yet contracted
                if ring.get("kron") is not None:
                    A = "kron" if zb > new_lo else None    # (nothing to produce: the PtAP forms the entries itself)
                else:
                    A = a_rows(new_lo * pf, zb * pf) if zb > new_lo else None
                ring["new"] = (new_lo, zb)
                b = b_rows(S["a_rows"][0], S["a_rows"][1]) if with_rhs else None
            else:
                A = a_rows(S["a_rows"][0], S["a_rows"][1])
                b = b_rows(S["a_rows"][0], S["a_rows"][1]) if with_rhs else None
            tick("input", t0)
            t0 = time.perf_counter()
            if use_factored:
                try:
                    kblk = self._factored_slab(A, S, ka, kb, zero_dofs if fold is None else None, diag, ring,
                                               builder if os.environ.get("TIGAR_SLAB_APPEND", "1") != "0" else None)
                except _TensorDeclined:
                    # A does not carry the element-coupling pattern (found while it was read): start over with
                    # the general stages (nothing of this call is kept)
                    self._tensor_declined = True
                    del A, b, builder, ring
                    return self.assemble(a_rows, b_rows, zero_dofs, diag, timers, a_factors)
                plan = None
            else:
                # (streamed in sub-slabs the operand rows of neighbouring blocks overlap: the fused kernel, which forms only
                #  what a row of K needs, is the faster one there -- 10.7 against 15.0 s at cfg3; one block: the library's rule)
                old_pref = dev.ptap_prefer(2) if nslabs > 1 else None
                try:
                    try:
                        plan = dev.ptap_symbolic(A, M, MT, S["a_rows"][0], S["m_rows"][0], S["dofs"][0])
                        kblk = dev.ptap_numeric(plan, A, M, MT, zero_dofs, diag)
                    except dev.TigarHipError as e:
                        if "does not cover" not in str(e):
                            raise
                        # columns beyond the FE rows the slab's elements couple to (a coupling added by hand): all rows of M
                        del M
                        M = dev.extract_csr_tensor(sp1, axes, 0, self.ncp, self.eps, 0, self.n_fe)
                        plan = dev.ptap_symbolic(A, M, MT, S["a_rows"][0], 0, S["dofs"][0])
                        kblk = dev.ptap_numeric(plan, A, M, MT, zero_dofs, diag)
                finally:
                    if old_pref is not None:
                        dev.ptap_prefer(old_pref)
            tick("ptap", t0)
            t0 = time.perf_counter()
            if kblk is True:
                pass                 # the last stage appended its rows to the builder itself
            elif nslabs == 1:
                k_blocks.append(kblk)
            else:
                if builder is None:
                    # capacity from the first slab's density (rows near the patch boundary are
                    # sparser than interior ones, hence the margin); grows if short
                    # from the densest (last = most interior) dof plane of the first slab: rows near
                    # the patch boundary are sparser, so the slab average underestimates and the
                    # builder would have to grow (a second 75 GB allocation and copy at cfg3)
                    pd = self.layout.plane_dofs if fold is None else int(np.prod(fold.ncp[:-1]))
                    nr = kblk.shape[0]
                    last = kblk.nnz - kblk.rowptr_at(nr - pd) if nr >= pd else kblk.nnz
                    est = int(max(kblk.nnz / max(1, kb - ka), last) * (self.k1 - self.k0) * 1.01) + 1024
                    builder = dev.CSRBuilder((self.k1 - self.k0) * pd, self.ncp if fold is None else pd * fold.ncp[-1], est)
                builder.append(kblk)
            del kblk
            tick("stack", t0)
            t0 = time.perf_counter()
            if with_rhs:
                if tensor_mtb:
                    y = self._mtb_tensor(b, S, ka, kb)
                else:
                    y = MT.mult_offset(b, S["a_rows"][0])
                y.zero_entries(zero_dofs, S["dofs"][0])
                rhs_parts.append(y)
                tick("mtb", t0)
            del A, M, MT, b, plan
        t0 = time.perf_counter()
        K = k_blocks[0] if builder is None else builder.finish()
        if fold is not None:
            K = fold.fold(K, zero_dofs, diag, planes=(self.k0, self.k1))
        rhs = None
        if with_rhs:
            rhs = rhs_parts[0] if len(rhs_parts) == 1 else dev.vec_concat(rhs_parts)
        tick("stack", t0)
        return K, rhs

    def elem_layers(self):
        """element layers [e0, e1) of the slab direction whose cells carry functions of this rank's dof planes, and the number
        of layers of the direction"""
        lay = self.layout
        if self.k1 <= self.k0:
            return 0, 0, lay.nel
        za, zb = lay.fe_planes_of_dofs(self.k0, self.k1)          # FE planes [za, zb) in the (closed) supports
        q = lay.q
        return max(0, za // q), min(lay.nel, -(-(zb - 1) // q)), lay.nel

    def _assemble_by_elements(self, a_rows, b_rows, zero_dofs, diag, timers):
        """K_loc and (M^T b)_loc with M and A used as GENERAL sparse matrices (tIGAr/common.py:1194-1200): the element split
        (``elemptap.ElementChunk``) over chunks of element layers of the slab direction, worked off from the bottom up
        (``assemble_blocks_by_elements``), MatZeroRowsColumns on the finished rows.  Returns None when the FE space or the matrix
        does not qualify (the row-wise stages take over)."""
        import time
        dev = self.dev
        out = self.assemble_blocks_by_elements([a_rows], timers)
        if out is None:
            return None
        K = out[0]
        zero_dofs = np.asarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        if zero_dofs.size and K.shape[0]:
            t0 = time.perf_counter()
            K.zero_rows_cols(zero_dofs, diag, self.mine["dofs"][0])
            if timers is not None:
                dev.sync()
                timers["stack"] = timers.get("stack", 0.0) + time.perf_counter() - t0
        rhs = self.assemble_vector(b_rows, zero_dofs, timers) if b_rows is not None else None
        return K, rhs

    def assemble_blocks_by_elements(self, producers, timers=None):
        """Rows of this rank of M^T A_j M for SEVERAL FE matrices A_j on the scalar space (``producers[j](r0, r1)`` = FE rows
        [r0, r1) of A_j as a DeviceCSR with global columns) -- the field blocks of a space with several fields on one basis.
        Chunks of element layers from the bottom up: a chunk sees the FE rows of its own layers (M materialised for them, A_j
        from the producers), lists the layer below it for the ownership rule, and its PLAN (function lists, incidence, pattern
        and places of K: the costly half) is built once and serves every A_j; dof planes whose cells all lie in finished chunks
        are final and go to the results, the top planes of a chunk are carried and added to the next chunk's rows.  Several
        ranks: every rank works off the layers in the support of ITS dof planes (the layers at a rank boundary twice: no
        exchange).  No boundary conditions.  Returns the list of K_j, or None when the FE space or one of the matrices does not
        qualify (the caller takes the row-wise stages)."""
        import time
        dev, lay, grid = self.dev, self.layout, self.grid
        from .elemptap import CellNodes, ElementChunk
        d, q = grid.dim(), int(grid.degree)
        if getattr(grid, "dg", False) or q < 1 or (q + 1) ** d > 125 or lay.q != q:
            return None
        t = timers if timers is not None else {}

        def tick(name, t0):
            dev.sync()
            t[name] = t.get(name, 0.0) + time.perf_counter() - t0

        g0, g1 = self.mine["dofs"]
        ncols = self.ncp
        import scipy.sparse as sp
        if g1 <= g0:
            return [dev.DeviceCSR.from_scipy(sp.csr_matrix((0, ncols))) for _ in producers]
        e_lo, e_hi, nel_z = self.elem_layers()
        nn = list(grid.shape())
        nel_other = [(n - 1) // q for n in nn[:-1]]
        cells_per_layer = int(np.prod(nel_other, dtype=np.int64)) if nel_other else 1
        pf, pd = lay.plane_fe, lay.plane_dofs
        L = self._elem_chunk_layers(d, q, cells_per_layer, pf, pd, e_hi - e_lo, len(producers))
        sp1, axes = self.basis.splines, grid.axes
        acc = [_ChunkRows(dev, g0, g1, ncols, pd) for _ in producers]
        e0 = e_lo
        while e0 < e_hi:
            e1 = min(e_hi, e0 + L)
            t0 = time.perf_counter()
            f0 = max(e0 - 1, 0)
            cells = CellNodes.from_grid(grid, [0] * (d - 1) + [f0], nel_other + [e1])
            r0, r1 = e0 * q * pf, (e1 * q + 1) * pf
            M = dev.extract_csr_tensor(sp1, axes, 0, self.ncp, self.eps, r0, r1)
            tick("extract", t0)
            t0 = time.perf_counter()
            try:
                chunk = ElementChunk(cells, M, r0, own=((e0 - f0) * cells_per_layer, (e1 - f0) * cells_per_layer))
            except ValueError:
                return None
            d0, d1 = chunk.dofs
            # dof planes whose cells all lie below the top of this chunk are complete
            done_planes = int(np.searchsorted(lay.sup_hi, e1 * q + 1, side="right")) if e1 < nel_z else lay.ncp
            tick("ptap", t0)
            for j, a_rows in enumerate(producers):
                t0 = time.perf_counter()
                A = a_rows(r0, r1)
                tick("input", t0)
                t0 = time.perf_counter()
                Kc = chunk.ptap(A, r0, (r0, r1 if e1 == nel_z else e1 * q * pf))
                del A
                if Kc is None:
                    return None                  # an entry between nodes of no common cell: the row-wise stages
                tick("ptap", t0)
                t0 = time.perf_counter()
                acc[j].add_chunk(Kc, d0, d1, done_planes * pd)
                del Kc
                tick("stack", t0)
            del chunk, M, cells
            if os.environ.get("TIGAR_DEBUG"):
                import sys
                sys.stderr.write("[tigar] element chunk layers %d..%d of %d..%d: cumulative %s\n"
                                 % (e0, e1, e_lo, e_hi, {k_: round(v, 3) for k_, v in t.items()}))
            e0 = e1
        t0 = time.perf_counter()
        out = [a.finish() for a in acc]
        tick("stack", t0)
        return out

    def _elem_chunk_layers(self, d, q, cells_per_layer, pf, pd, nlayers, nblocks=1):
        """element layers per chunk of the element-split stage: blocks + places + lists of the layer's cells, q node planes of A
        and M, the chunk's rows of K (and their copies while chunks are added) in about 40 % of the free device memory"""
        if os.environ.get("TIGAR_ELEM_LAYERS"):
            return max(1, int(os.environ["TIGAR_ELEM_LAYERS"]))
        dev = self.dev
        free_b = dev.mem_info()[0] + dev.pool_stats()[0]
        if self.world > 1:
            try:
                local = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
                free_b //= max(1, -(-local // max(1, dev.device_count())))
            except Exception:
                pass
        b = (q + 1) ** d
        per_cell = b * b * 10 + b * 64 + 128 * 4 + b * 8
        per_node = ((2 * q + 1) ** d) * 0.55 * 12.0 + ((q + 1) ** d) * 12.0 * 1.2
        per_dof = ((2 * q + 1) ** d) * 12.0 * 3.0
        per_layer = cells_per_layer * per_cell + q * pf * per_node + pd * per_dof
        # (several blocks on one plan: the carried planes of every block stay, and so do the blocks' finished rows -- K of the
        #  whole rank per block; the caller's sub-slab sizing has subtracted those already where it knows them)
        fixed = pf * per_node + nblocks * (q + 1) * pd * per_dof
        return int(max(1, min(nlayers, (0.4 * free_b - fixed) // per_layer)))

    def _assemble_pair(self, a_rows, col, timers=None, a_factors=None):
        """rows [k0, k1) of K_fg = M_f^T A_fg M_g, f = this engine's basis, g = ``col``: the same sub-slab pipeline with the
        pair plan of the tensor-pattern walks (x / y passes once per FE plane, kept in the ring; the z pass appends the
        rows at closed-form positions); no boundary conditions (they belong to the assembled matrix)"""
        import time
        from .tensorptap import TensorPtAP
        dev = self.dev
        t = timers if timers is not None else {}
        tplan = TensorPtAP.for_pair(self.kx, col)
        if tplan is None:
            raise NotImplementedError("the streamed / multi-GPU path for fields on different bases needs tensor-product B-spline "
                                      "fields of degree <= 3 on one 3-D node grid (the line walks)")
        packed = tplan.pack_kron_factors(a_factors) if (a_factors is not None and os.environ.get("TIGAR_PTAP_FUSED", "1") != "0") else None
        subs = self.sub_slabs()
        pd = self.layout.plane_dofs
        ncols = int(np.prod(col.ncp, dtype=np.int64))
        nrows = (self.k1 - self.k0) * pd
        if not subs:
            import scipy.sparse as sp
            return dev.DeviceCSR.from_scipy(sp.csr_matrix((0, ncols))), None
        builder = dev.CSRBuilder(nrows, ncols, tplan.k_nnz(self.k0, self.k1)) if len(subs) > 1 else None
        pieces, hi, out = [], 0, None
        pf = self.layout.plane_fe
        for (ka, kb) in subs:
            S = self.layout.slab(ka, kb)
            za, zb = S["a_rows"][0] // pf, S["a_rows"][1] // pf
            new_lo = max(za, hi)
            t0 = time.perf_counter()
            if zb > new_lo:
                if packed is not None:
                    piece = tplan.planes_kron(packed, new_lo, zb)
                    if piece is None:
                        packed = None
                if packed is None:
                    A = a_rows(new_lo * pf, zb * pf)
                    dev.sync()
                    t["input"] = t.get("input", 0.0) + time.perf_counter() - t0
                    piece = tplan.planes(A, new_lo * pf, new_lo, zb)
                    del A
                if piece is None:
                    raise NotImplementedError("the streamed / multi-GPU path for fields on different bases needs FE matrices "
                                              "on the element-coupling pattern of the node grid")
                pieces.append((new_lo, zb, piece))
                hi = zb
            pieces = [pc for pc in pieces if pc[1] > za]
            out = tplan.zstage([pc[2] for pc in pieces], ka, kb, None, 1.0, append_to=builder)
            dev.sync()
            t["ptap"] = t.get("ptap", 0.0) + time.perf_counter() - t0
        return (builder.finish() if builder is not None else out), None

    def assemble_matrix(self, a_rows, zero_dofs, diag=1.0, timers=None, a_factors=None):
        """K_loc = rows of M^T A M owned by this rank (extractMatrix, tIGAr/common.py:1176-1204)."""
        return self.assemble(a_rows, None, zero_dofs, diag, timers, a_factors)[0]

    def assemble_vector(self, b_rows, zero_dofs=None, timers=None):
        """(M^T b)_loc with the boundary entries zeroed (extractVector, tIGAr/common.py:1142-1160): one
        sum-factorised pass over the FE rows in the support of this rank's dofs when M is exactly the
        Kronecker product, explicit M^T rows sub-slab by sub-slab otherwise.  ``b_rows(r0, r1)`` returns
        the FE entries [r0, r1) as a DeviceVector."""
        import time
        dev = self.dev
        t0 = time.perf_counter()
        zero_dofs = np.asarray(zero_dofs if zero_dofs is not None else [], dtype=np.int32)
        if self.kron_exact:
            S = self.mine
            y = self._mtb_tensor(b_rows(S["a_rows"][0], S["a_rows"][1]), S, self.k0, self.k1)
        else:
            parts = []
            for (ka, kb) in self.sub_slabs():
                S = self.layout.slab(ka, kb)
                MT = dev.extract_csr_tensor_t(self.basis.splines, self.grid.axes, 0, self.n_fe, self.eps,
                                              S["dofs"][0], S["dofs"][1])
                parts.append(MT.mult_offset(b_rows(S["a_rows"][0], S["a_rows"][1]), S["a_rows"][0]))
                del MT
            y = parts[0] if len(parts) == 1 else dev.vec_concat(parts)
        y.zero_entries(zero_dofs, self.mine["dofs"][0])
        if timers is not None:
            dev.sync()
            timers["mtb"] = timers.get("mtb", 0.0) + time.perf_counter() - t0
        return y

    def _factored_slab(self, A_new, S, ka, kb, zero_dofs, diag, ring, builder=None):
        """Sum-factorised K rows of dof planes [ka,kb).  The plane-local stages (all direction
        groups but the last) are applied once per FE plane: their results are kept in ``ring``
        and shared by neighbouring sub-slabs (their supports overlap by ~p*p planes)."""
        from .kronptap import contract
        dev, kx, lay = self.dev, self.kx, self.layout
        pf = lay.plane_fe
        za, zb = S["a_rows"][0] // pf, S["a_rows"][1] // pf
        new_lo, new_hi = ring["new"]
        tplan = ring.get("tensor")
        if tplan is not None:
            if A_new is not None:
                if isinstance(A_new, str):                 # "kron": the matrix is a Kronecker sum, formed inside the x pass
                    piece = tplan.planes_kron(ring["kron"], new_lo, new_hi)
                    if piece is None:
                        ring["kron"] = None
                else:
                    piece = tplan.planes(A_new, new_lo * pf, new_lo, new_hi)
                if piece is None:
                    raise _TensorDeclined()
                ring["pieces"].append((new_lo, new_hi, piece))
                ring["hi"] = new_hi
            ring["pieces"] = [pc for pc in ring["pieces"] if pc[1] > za]
            return tplan.zstage([pc[2] for pc in ring["pieces"]], ka, kb, zero_dofs, diag, append_to=builder)
        if A_new is not None:
            cur, done = A_new, set()
            ca, cb = lay.fe_planes_coupled(new_lo, new_hi)
            if not kx.box_kernels_safe(A_new, new_lo, new_hi):
                ring["nobox"] = True           # (repeated knots + a matrix off the element-coupling pattern: see KronExtraction)
            for group in self.groups[:-1]:
                after = done | set(group)
                pl_out = kx.plane(after)
                cur = contract(kx, cur, done, group, (new_lo, new_hi), (ca, cb), (new_lo * pl_out, new_hi * pl_out),
                               intermediate=True, box=not ring.get("nobox", False))
                done = after
            ring["pieces"].append((new_lo, new_hi, cur))
            ring["hi"] = new_hi
        ring["pieces"] = [pc for pc in ring["pieces"] if pc[1] > za]     # drop planes below the slab
        lo = ring["pieces"][0][0]
        blocks = [pc[2] for pc in ring["pieces"]]
        # the last stage reads the pieces in place (row tables only; TIGAR_RING_COPY=1: stacked copy)
        if len(blocks) == 1:
            cur = blocks[0]
        elif os.environ.get("TIGAR_RING_COPY") == "1":
            cur = dev.csr_vstack(blocks)
        else:
            cur = dev.csr_vstack_view(blocks)
        done = set(sum(self.groups[:-1], []))
        ca, cb = lay.fe_planes_coupled(lo, zb)
        pl_out = kx.plane(done | set(self.groups[-1]))
        return contract(kx, cur, done, self.groups[-1], (lo, zb), (ca, cb), (ka * pl_out, kb * pl_out), zero_dofs, diag,
                        append_to=builder, box=not ring.get("nobox", False))

    def solve(self, K, rhs, method="cg", pc="jacobi", rtol=1e-6, atol=1e-15, maxit=10000, restart=30, x0=None):
        dev = self.dev
        U = dev.DeviceVector(K.shape[0]) if x0 is None else x0
        its, res, status = dev.krylov_solve(K, rhs, U, method, pc, rtol, atol, maxit, restart,
                                            self.comm if self.world > 1 else None,
                                            nonzero_initial_guess=x0 is not None)
        return U, its, res, status

    def _mtb_tensor(self, b, S, ka, kb):
        """(M^T b) of the dof planes [ka,kb) from the FE rows S["a_rows"] of b by sum factorisation:
        M^T = M_z^T (x) M_y^T (x) M_x^T applied direction by direction (tIGAr/common.py:97-109 without
        forming M^T; valid because M is exactly the Kronecker product, ``self.kron_exact``)."""
        dev, kx, d = self.dev, self.kx, self.kx.d
        pf = self.layout.plane_fe
        za, zb = S["a_rows"][0] // pf, S["a_rows"][1] // pf
        dims = list(kx.nfe[:-1]) + [zb - za]
        t = b
        for k in range(d - 1):
            t = dev.tensor_apply_1d(t, dims, k, kx.M1T[k])
            dims[k] = kx.ncp[k]
        MzT = kx.M1T[-1][ka:kb]
        return dev.tensor_apply_1d(t, dims, d - 1, MzT, col_shift=za)

    def _prolong_tensor(self, x, dof_plane0):
        """u = M U on the FE planes this rank owns, U given on the dof planes starting at
        ``dof_plane0`` (own planes + halo), direction by direction."""
        dev, kx, d = self.dev, self.kx, self.kx.d
        pf = self.layout.plane_fe
        fa, fb = self.mine["u_rows"][0] // pf, self.mine["u_rows"][1] // pf
        nk = x.size() // self.layout.plane_dofs
        dims = list(kx.ncp[:-1]) + [nk]
        t = x
        for k in range(d - 1):
            t = dev.tensor_apply_1d(t, dims, k, kx.M1[k])
            dims[k] = kx.nfe[k]
        Mz = kx.M1[-1][fa:fb]
        return dev.tensor_apply_1d(t, dims, d - 1, Mz, col_shift=dof_plane0)

    def prolong(self, U):
        """u rows owned by this rank: u = M_own * U (tIGAr/common.py:1259), U with its halo.
        Matrix-free: the rows of M are evaluated and contracted with U on the fly."""
        dev = self.dev
        r0, r1 = self.mine["u_rows"]
        if self.world > 1:
            x = self.comm.halo_extend(U)
            x_col0 = self.mine["dofs"][0] - self.mine["halo"][0]
        else:
            x, x_col0 = U, 0
        if self.kron_exact:
            return self._prolong_tensor(x, x_col0 // self.layout.plane_dofs)
        return dev.extract_apply_tensor(self.basis.splines, self.grid.axes, 0, self.eps, x, x_col0, r0, r1)


def _block_is_empty(a_block, f, g, ar, plane_fe, fac):
    """Are fields f and g uncoupled on the FE rows [ar[0], ar[1]) of a rank?  The producer says so itself (None), or three node
    planes of the block -- the first, the middle, the last of the rows -- hold no entry.  (Until round 5 the whole row block of
    the rank was assembled for this question and thrown away: twice the FE assembly, and in ONE piece the rows the sub-slab
    pipeline exists to avoid holding at once -- ADVICE r5.  A block of an assembled form is structurally empty or it has
    entries on every node plane; couplings added by hand come as explicit matrices, whose blocks are cut out, not assembled.)"""
    r0, r1 = int(ar[0]), int(ar[1])
    if r1 <= r0:
        return True
    if fac is not None:
        return a_block(f, g, r0, min(r0 + 1, r1)) is None
    pf = max(1, int(plane_fe))
    starts = sorted({r0, r0 + ((r1 - r0) // (2 * pf)) * pf, max(r0, r1 - pf)})
    for q0 in starts:
        probe = a_block(f, g, q0, min(q0 + pf, r1))
        if probe is None:
            return True
        if probe.nnz:
            return False
    return True


class FieldSlabPath(object):
    """The z-slab path for ``nfields`` fields on ONE tensor basis (M = diag(M_s, ..., M_s); the reference numbers such dofs
    field after field, tIGAr/common.py:242-252).  A rank owns the dof planes [k0, k1) of EVERY field, so that its rows of
    K form one contiguous block and the Krylov halo is p planes on either side, the distributed numbering interleaves
    the fields plane by plane:

        new index of (field f, plane k, in-plane index ij)  =  k * nF * pd  +  f * pd  +  ij      (pd = dofs per plane)

    -- a renumbering of the IGA dofs for parallel runs as the reference's ``generatePermutation`` is one
    (``local_dof_indices()`` gives the reference index of every local entry).  Block (f, g) of the product is M_s^T A_fg M_s:
    the scalar ``SlabHotPath`` computes its rows for the rank's planes (tensor-pattern passes, fused Kronecker forms,
    general stages -- whatever the block qualifies for), the blocks are put together, rows and columns brought into the
    interleaved order, and MatZeroRowsColumns is applied to the whole (tIGAr/common.py:1194-1200)."""

    def __init__(self, basis, grid, nfields, rank=0, world=1, comm=None, sub_planes="auto", eps=1e-15, kx=None):
        from . import device as dev
        self.dev = dev
        self.nF = int(nfields)
        # the scalar engine never talks to the communicator: the slab of the Krylov vectors is set here, for all fields
        self.scalar = SlabHotPath(basis, grid, rank, world, None, sub_planes, eps, kx=kx, resident_blocks=2 * self.nF * self.nF)
        self.rank, self.world, self.comm = rank, world, comm
        S = self.scalar
        self.layout = S.layout
        self.pd = S.layout.plane_dofs
        self.k0, self.k1 = S.k0, S.k1
        self.ncp1 = S.ncp                       # dofs of one field
        self.nfe1 = S.n_fe
        self.ncp = self.nF * self.ncp1
        self.sub_planes = S.sub_planes
        nF, pd = self.nF, self.pd
        hl, hh = S.mine["halo"][0] // pd, S.mine["halo"][1] // pd
        self.halo_planes = (hl, hh)
        ur = S.mine["u_rows"]
        self.mine = {"dofs": (self.k0 * nF * pd, self.k1 * nF * pd),
                     "halo": (hl * nF * pd, hh * nF * pd),
                     "u_rows": [(f * self.nfe1 + ur[0], f * self.nfe1 + ur[1]) for f in range(nF)]}
        if comm is not None and world > 1:
            comm.set_slab(self.mine["dofs"][0], self.mine["dofs"][1], self.mine["halo"][0], self.mine["halo"][1], self.ncp)

    # ---- numbering ------------------------------------------------------------------------------------------
    def new_of_old(self):
        """distributed index of every reference (field-major) dof"""
        nF, pd, n1 = self.nF, self.pd, self.ncp1
        old = np.arange(self.ncp, dtype=np.int64)
        f, s = old // n1, old % n1
        return (s // pd) * (nF * pd) + f * pd + (s % pd)

    def local_dof_indices(self):
        """reference (field-major) index of every entry of this rank's vectors / rows of K, in local order"""
        nF, pd, n1 = self.nF, self.pd, self.ncp1
        loc = np.arange(self.mine["dofs"][0], self.mine["dofs"][1], dtype=np.int64)
        k, r = loc // (nF * pd), loc % (nF * pd)
        return (r // pd) * n1 + k * pd + (r % pd)

    def _interleave_vec(self, parts):
        """[field][plane-major local] -> local vector in the interleaved order"""
        dev, nF, pd = self.dev, self.nF, self.pd
        nk = self.k1 - self.k0
        out = dev.DeviceVector(nk * nF * pd)
        for k in range(nk):
            for f in range(nF):
                dev.vec_copy_range(out, (k * nF + f) * pd, parts[f], k * pd, pd)
        return out

    # ---- the path ---------------------------------------------------------------------------------------------
    def assemble_matrix(self, a_block, zero_dofs, diag=1.0, timers=None, block_factors=None):
        """``a_block(f, g, r0, r1)``: rows [r0, r1) of block (f, g) of the FE matrix as a DeviceCSR with the columns of
        ONE field (0 .. nfe-1), or None when the fields are not coupled; ``block_factors[f][g]``: the block as a
        Kronecker sum of 1-D matrices (fused into the first pass where the patch qualifies)."""
        dev, S, nF, pd = self.dev, self.scalar, self.nF, self.pd
        nloc1 = (self.k1 - self.k0) * pd
        import scipy.sparse as sp
        empty = [[_block_is_empty(a_block, f, g, S.mine["a_rows"], S.layout.plane_fe,
                                  block_factors[f][g] if block_factors is not None else None) for g in range(nF)] for f in range(nF)]
        together = None
        if not S.factored and os.environ.get("TIGAR_PTAP_ELEMENTS", "1") != "0":
            # nothing assumed about M or A: ONE pass over the element chunks serves all coupled blocks (a chunk's plan is the
            # costly half and depends on M only)
            pairs = [(f, g) for f in range(nF) for g in range(nF) if not empty[f][g]]
            outs = S.assemble_blocks_by_elements([(lambda r0, r1, f=f, g=g: a_block(f, g, r0, r1)) for f, g in pairs], timers) \
                if pairs else []
            if outs is not None:
                together = dict(zip(pairs, outs))
        blocks = []
        for f in range(nF):
            row = []
            for g in range(nF):
                fac = block_factors[f][g] if block_factors is not None else None
                if empty[f][g]:
                    # fields f and g are not coupled (on this rank's rows): no entries in this block of the product
                    row.append(dev.DeviceCSR.from_scipy(sp.csr_matrix((nloc1, self.ncp1))))
                    continue
                if together is not None:
                    row.append(together[(f, g)])
                    continue
                S._tensor_declined = False          # (every block is judged on its own pattern)
                Kfg = S.assemble(lambda r0, r1, f=f, g=g: a_block(f, g, r0, r1), None, None, 1.0, timers, fac)[0]
                row.append(Kfg)
            blocks.append(row)
        del together
        K = dev.csr_from_blocks(blocks)            # rows (f, k - k0, ij) local, columns (g, k', ij') field-major
        del blocks
        nk = self.k1 - self.k0
        # rows into (k, f, ij) order
        l = np.arange(nk * nF * pd, dtype=np.int64)
        k, r = l // (nF * pd), l % (nF * pd)
        K = K.gather_rows((r // pd) * (nk * pd) + k * pd + (r % pd))
        n2o = self.new_of_old()
        K = K.permute_columns(n2o)
        if zero_dofs is not None and len(zero_dofs):
            K.zero_rows_cols(n2o[np.asarray(zero_dofs, dtype=np.int64)].astype(np.int32), diag, self.mine["dofs"][0])
        return K

    def assemble_vector(self, b_rows, zero_dofs=None, timers=None):
        dev, S, nF = self.dev, self.scalar, self.nF
        parts = [S.assemble_vector(lambda r0, r1, f=f: b_rows(f * self.nfe1 + r0, f * self.nfe1 + r1), None, timers)
                 for f in range(nF)]
        y = self._interleave_vec(parts)
        if zero_dofs is not None and len(zero_dofs):
            y.zero_entries(self.new_of_old()[np.asarray(zero_dofs, dtype=np.int64)].astype(np.int32), self.mine["dofs"][0])
        return y

    def prolong(self, U):
        """FE rows of u = M U this rank owns, field after field (``mine["u_rows"]``), from the local U (interleaved)"""
        dev, S, nF, pd = self.dev, self.scalar, self.nF, self.pd
        hl, hh = self.halo_planes
        if self.world > 1:
            x = self.comm.halo_extend(U)
            plane0 = self.k0 - hl
        else:
            x, plane0 = U, self.k0
        nplanes = x.size() // (nF * pd)
        out = []
        for f in range(nF):
            xf = dev.DeviceVector(nplanes * pd)
            for k in range(nplanes):
                dev.vec_copy_range(xf, k * pd, x, (k * nF + f) * pd, pd)
            out.append(S._prolong_tensor(xf, plane0))
        return dev.vec_concat(out)


class FieldListSlabPath(object):
    """The z-slab path for fields on DIFFERENT tensor bases over one FE node grid -- ``FieldListSpline``, the components of a
    compatible B-spline (tIGAr/common.py:1949-1970, tIGAr/compatibleSplines.py:21-101; the spaces of the reference's
    Krylov + MPI demos, demos/taylor-green/taylor-green-3d.py:42-90).  M = diag(M_0, ..., M_{nF-1}); field f has its own number
    of dof planes nk_f (elements + degree in the slab direction) and of dofs per plane pd_f.

    One split of the PLANE INDEX k serves all fields: a rank owns the planes [K0, K1) of every field that has them.  The
    distributed numbering interleaves the fields plane by plane,

        new index of (field f, plane k, in-plane index ij)  =  off[k] + sum_{f' < f, k < nk_f'} pd_f' + ij,

    so that a rank's rows of K are one contiguous block and the Krylov halo is a contiguous run of max-degree planes on either
    side (functions (f, i) and (g, j) couple for j in [i - p_f, i + p_g], in their own plane indices) -- a renumbering of the
    IGA dofs for parallel runs as the reference's ``generatePermutation`` is one; ``local_dof_indices()`` names the reference
    (field-after-field) index of every local entry.  Block (f, g) = M_f^T A_fg M_g comes from the scalar engine of field f
    with the column side of field g (``SlabHotPath.assemble(col=...)``: the line walks with separate row / column weights)."""

    def __init__(self, kxs, rank=0, world=1, comm=None, sub_planes="auto", eps=1e-15):
        from . import device as dev
        self.dev = dev
        self.kxs = list(kxs)
        self.nF = len(self.kxs)
        self.rank, self.world, self.comm = rank, world, comm
        g0 = self.kxs[0].grid
        if any(kx.d != self.kxs[0].d or any(not np.array_equal(a, b) for a, b in zip(kx.grid.axes, g0.axes)) for kx in self.kxs):
            raise NotImplementedError("the streamed / multi-GPU path needs all fields on one FE node grid")
        self.nk = [int(kx.ncp[-1]) for kx in self.kxs]
        self.pd = [int(np.prod(kx.ncp[:-1], dtype=np.int64)) if kx.d > 1 else 1 for kx in self.kxs]
        self.ncp_f = [nk * pd for nk, pd in zip(self.nk, self.pd)]
        self.nfe1 = int(np.prod(self.kxs[0].nfe, dtype=np.int64))
        self.ncp = int(sum(self.ncp_f))
        self.Kmax = max(self.nk)
        self.K0, self.K1 = split_range(self.Kmax, world)[rank]
        nF = self.nF
        # off[k]: first new index of plane k; present[k][f]
        self.width = np.array([[self.pd[f] if k < self.nk[f] else 0 for f in range(nF)] for k in range(self.Kmax)], dtype=np.int64)
        self.off = np.concatenate([[0], np.cumsum(self.width.sum(axis=1))])
        self.engines = []
        for f, kx in enumerate(self.kxs):
            k0, k1 = min(self.K0, self.nk[f]), min(self.K1, self.nk[f])
            self.engines.append(SlabHotPath(kx.basis, kx.grid, rank, world, None, sub_planes, eps, kx=kx, planes=(k0, k1),
                                            resident_blocks=2 * self.nF * self.nF))
        self.sub_planes = self.engines[0].sub_planes
        H = max(max(s1.p for s1 in kx.basis.splines[-1:]) for kx in self.kxs)
        lo, hi = max(0, self.K0 - H), min(self.Kmax, self.K1 + H)
        self.halo_planes = (self.K0 - lo, hi - self.K1)
        self.mine = {"dofs": (int(self.off[self.K0]), int(self.off[self.K1])),
                     "halo": (int(self.off[self.K0] - self.off[lo]), int(self.off[hi] - self.off[self.K1])),
                     "u_rows": [(f * self.nfe1 + e.mine["u_rows"][0], f * self.nfe1 + e.mine["u_rows"][1])
                                for f, e in enumerate(self.engines)]}
        if comm is not None and world > 1:
            comm.set_slab(self.mine["dofs"][0], self.mine["dofs"][1], self.mine["halo"][0], self.mine["halo"][1], self.ncp)

    # ---- numbering ------------------------------------------------------------------------------------------
    def _field_offsets_old(self):
        return np.concatenate([[0], np.cumsum(self.ncp_f)])

    def new_of_old(self):
        """distributed index of every reference (field-after-field) dof"""
        out = np.empty(self.ncp, dtype=np.int64)
        fo = self._field_offsets_old()
        for f in range(self.nF):
            k = np.repeat(np.arange(self.nk[f], dtype=np.int64), self.pd[f])
            ij = np.tile(np.arange(self.pd[f], dtype=np.int64), self.nk[f])
            out[fo[f]:fo[f + 1]] = self.off[k] + self.width[k, :f].sum(axis=1) + ij
        return out

    def local_dof_indices(self):
        """reference (field-after-field) index of every entry of this rank's vectors / rows of K, in local order"""
        n2o = np.empty(self.ncp, dtype=np.int64)
        n2o[self.new_of_old()] = np.arange(self.ncp, dtype=np.int64)
        return n2o[self.mine["dofs"][0]:self.mine["dofs"][1]]

    def _local_rows_field_major(self):
        """for every local entry in the interleaved order: its position in the field-major local stacking
        [field 0: planes k0_0..k1_0][field 1: ...] that the blocks come in"""
        pos, base = [], 0
        starts = []
        for f, e in enumerate(self.engines):
            starts.append(base)
            base += (e.k1 - e.k0) * self.pd[f]
        for k in range(self.K0, self.K1):
            for f, e in enumerate(self.engines):
                if k < self.nk[f]:
                    pos.append(starts[f] + (k - e.k0) * self.pd[f] + np.arange(self.pd[f], dtype=np.int64))
        return np.concatenate(pos) if pos else np.zeros(0, dtype=np.int64)

    def _interleave_vec(self, parts):
        dev = self.dev
        out = dev.DeviceVector(self.mine["dofs"][1] - self.mine["dofs"][0])
        at = 0
        for k in range(self.K0, self.K1):
            for f, e in enumerate(self.engines):
                if k < self.nk[f]:
                    dev.vec_copy_range(out, at, parts[f], (k - e.k0) * self.pd[f], self.pd[f])
                    at += self.pd[f]
        return out

    # ---- the path ---------------------------------------------------------------------------------------------
    def assemble_matrix(self, a_block, zero_dofs, diag=1.0, timers=None, block_factors=None):
        """``a_block(f, g, r0, r1)``: rows [r0, r1) of block (f, g) of the FE matrix (FE rows of field f counted from 0) as a
        DeviceCSR with the columns of field g (0 .. nfe-1), or None when the fields are not coupled; ``block_factors[f][g]``:
        the block as a Kronecker sum of 1-D matrices (formed inside the first pass)."""
        import scipy.sparse as sp
        dev, nF = self.dev, self.nF
        blocks = []
        for f, e in enumerate(self.engines):
            row = []
            nloc = (e.k1 - e.k0) * self.pd[f]
            for g in range(nF):
                fac = block_factors[f][g] if block_factors is not None else None
                if e.k1 <= e.k0 or _block_is_empty(a_block, f, g, e.mine["a_rows"], e.layout.plane_fe, fac):
                    row.append(dev.DeviceCSR.from_scipy(sp.csr_matrix((nloc, self.ncp_f[g]))))
                    continue
                Kfg = e.assemble(lambda r0, r1, f=f, g=g: a_block(f, g, r0, r1), None, None, 1.0, timers, fac, col=self.kxs[g])[0]
                row.append(Kfg)
            blocks.append(row)
        K = dev.csr_from_blocks(blocks)            # rows field-major local, columns field-after-field (reference numbering)
        del blocks
        K = K.gather_rows(self._local_rows_field_major())
        n2o = self.new_of_old()
        K = K.permute_columns(n2o)
        if zero_dofs is not None and len(zero_dofs):
            K.zero_rows_cols(n2o[np.asarray(zero_dofs, dtype=np.int64)].astype(np.int32), diag, self.mine["dofs"][0])
        return K

    def assemble_vector(self, b_rows, zero_dofs=None, timers=None):
        parts = []
        for f, e in enumerate(self.engines):
            if e.k1 > e.k0:
                parts.append(e.assemble_vector(lambda r0, r1, f=f: b_rows(f * self.nfe1 + r0, f * self.nfe1 + r1), None, timers))
            else:
                parts.append(self.dev.DeviceVector(0))
        y = self._interleave_vec(parts)
        if zero_dofs is not None and len(zero_dofs):
            y.zero_entries(self.new_of_old()[np.asarray(zero_dofs, dtype=np.int64)].astype(np.int32), self.mine["dofs"][0])
        return y

    def prolong(self, U):
        """FE rows of u = M U this rank owns, field after field (``mine["u_rows"]``), from the local U (interleaved)"""
        dev = self.dev
        hl, hh = self.halo_planes
        if self.world > 1:
            x = self.comm.halo_extend(U)
            kk0, kk1 = self.K0 - hl, self.K1 + hh
        else:
            x, kk0, kk1 = U, self.K0, self.K1
        out = []
        for f, e in enumerate(self.engines):
            ka, kb = min(kk0, self.nk[f]), min(kk1, self.nk[f])
            xf = dev.DeviceVector(max(0, kb - ka) * self.pd[f])
            for k in range(ka, kb):
                src = int(self.off[k] - self.off[kk0] + self.width[k, :f].sum())
                dev.vec_copy_range(xf, (k - ka) * self.pd[f], x, src, self.pd[f])
            out.append(e._prolong_tensor(xf, ka))
        return dev.vec_concat(out)
