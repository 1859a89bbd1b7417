"""CPU: the per-lane code of the tensor-pattern PtAP kernels (tigar_amd/csrc/tg_tensor_body.h), executed lane by
lane by a host build of the same header (tests/emu/tensor_emu.cpp), against the oracle's M^T A M +
MatZeroRowsColumns on random values over the element-coupling pattern.  Pins the index arithmetic of the three
line walks (row-block addressing, rings of live output rows, sub-slab seams of the z pass, closed-form CSR
positions, fused boundary conditions) and the host-side structure checks of tigar_amd/tensorptap.py without a GPU.
The device build of the same code is covered by the -m gpu tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
c_f64p, c_i32p, c_i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


@pytest.fixture(scope="module")
def emu():
    build = os.path.join(HERE, "emu", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libtensor_emu.so")
    src = os.path.join(HERE, "emu", "tensor_emu.cpp")
    hdr = os.path.join(HERE, "..", "tigar_amd", "csrc", "tg_tensor_body.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", src, "-o", so])
    return C.CDLL(so)


def _p(a, t):
    return a.ctypes.data_as(t)


def _rn(p, a, nfe):
    return 2 * p + 1 if (a % p == 0 and 0 < a < nfe - 1) else p + 1


def _tables(p, nel):
    nfe, ncp = p * nel + 1, nel + p
    rps = np.zeros(nfe + 1, dtype=np.int32)
    for a in range(nfe):
        rps[a + 1] = rps[a] + _rn(p, a, nfe)
    kps = np.zeros(ncp + 1, dtype=np.int32)
    for i in range(ncp):
        kps[i + 1] = kps[i] + (min(ncp - 1, i + p) - max(0, i - p) + 1)
    return rps, kps


def _emulated_ptap(emu, p, nels, wls, A, plane_splits, dof_splits, zero_dofs, diag):
    """mirror of tg_tensor_planes / tg_tensor_zstage (csrc/tg_ptap_tensor.hip) on host arrays"""
    W = 2 * p + 1
    nfe = [p * n + 1 for n in nels]
    ncp = [n + p for n in nels]
    tabs = [_tables(p, n) for n in nels]
    A = sp.csr_matrix(A)
    A.sort_indices()
    rowptr, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)
    lines = [np.array([a for a in range(nfe[1]) if _rn(p, a, nfe[1]) == n1], dtype=np.int32) for n1 in (p + 1, W)]
    pieces = []
    for (z0, z1) in plane_splits:
        npl = z1 - z0
        pb1 = np.zeros(npl + 1, dtype=np.int64)
        pb2 = np.zeros(npl + 1, dtype=np.int64)
        pls = [[], []]
        for q in range(npl):
            n2 = _rn(p, z0 + q, nfe[2])
            pb1[q + 1] = pb1[q] + W * n2 * ncp[0] * int(tabs[1][0][nfe[1]])
            pb2[q + 1] = pb2[q] + W * W * n2 * ncp[0] * ncp[1]
            pls[0 if n2 == p + 1 else 1].append(z0 + q)
        b1 = np.full(int(pb1[npl]), np.nan)
        b2 = np.full(int(pb2[npl]), np.nan)
        # A slab: rows of the planes [z0, z1) only, as the streamed path hands them over
        r0, r1 = z0 * nfe[0] * nfe[1], z1 * nfe[0] * nfe[1]
        rp = (rowptr[r0:r1 + 1] - rowptr[r0]).copy()
        cs, vs = col[rowptr[r0]:rowptr[r1]].copy(), val[rowptr[r0]:rowptr[r1]].copy()
        for pc in range(2):
            if not pls[pc]:
                continue
            n2 = p + 1 if pc == 0 else W
            planes = np.array(pls[pc], dtype=np.int32)
            for lc in range(2):
                if not len(lines[lc]):
                    continue
                n1 = p + 1 if lc == 0 else W
                L = max(1, 64 // (n1 * n2))
                bad = emu.emu_x(p, _p(rp, c_i64p), _p(cs, c_i32p), _p(vs, c_f64p), z0, nels[0], nfe[1], nfe[2],
                                _p(wls[0], c_f64p), _p(tabs[0][0], c_i32p), _p(tabs[1][0], c_i32p), _p(tabs[2][0], c_i32p),
                                _p(lines[lc], c_i32p), len(lines[lc]), L, n1,
                                _p(planes, c_i32p), len(planes), n2, _p(b1, c_f64p), _p(pb1, c_i64p), z0,
                                C.c_int64(r1 - r0))
                if bad:
                    return None
            L = max(1, 64 // (W * n2))
            emu.emu_y(p, _p(b1, c_f64p), _p(pb1, c_i64p), _p(b2, c_f64p), _p(pb2, c_i64p), z0, nels[1],
                      _p(wls[1], c_f64p), _p(tabs[1][0], c_i32p), ncp[0], _p(planes, c_i32p), len(planes), n2, L)
        assert not np.any(np.isnan(b1)) and not np.any(np.isnan(b2))       # every block entry was written
        pieces.append((z0, z1, b2, pb2))
    ntot = ncp[0] * ncp[1] * ncp[2]
    mask = None
    if zero_dofs is not None and len(zero_dofs):
        mask = np.zeros(ntot, dtype=np.uint8)
        mask[np.asarray(zero_dofs)] = 1
    w01 = int(tabs[0][1][ncp[0]]) * int(tabs[1][1][ncp[1]])
    nnz = w01 * int(tabs[2][1][ncp[2]])
    kcol = np.full(nnz, -1, dtype=np.int32)
    kval = np.full(nnz, np.nan)
    krow = np.zeros(ntot + 1, dtype=np.int64)
    pd = ncp[0] * ncp[1]
    at = 0
    for (ka, kb) in dof_splits:
        e_begin, e_end = max(0, ka - p), min(nels[2], kb)
        plo, phi = (0 if e_begin == 0 else p * e_begin + 1), p * e_end
        ptrs = (c_f64p * (phi - plo + 1))()
        for r in range(plo, phi + 1):
            (z0, z1, b2, pb2) = [pc for pc in pieces if pc[0] <= r < pc[1]][0]
            ptrs[r - plo] = C.cast(b2.ctypes.data + 8 * int(pb2[r - z0]), c_f64p)
        cnt = w01 * int(tabs[2][1][kb] - tabs[2][1][ka])
        emu.emu_z(p, ptrs, plo, nels[2], _p(wls[2], c_f64p), _p(tabs[2][1], c_i32p), ncp[0], ncp[1],
                  _p(tabs[0][1], c_i32p), _p(tabs[1][1], c_i32p), ka, kb, max(1, 64 // (W * W)),
                  C.cast(kcol.ctypes.data + 4 * at, c_i32p), C.cast(kval.ctypes.data + 8 * at, c_f64p),
                  _p(mask, C.POINTER(C.c_uint8)) if mask is not None else None, C.c_double(diag),
                  C.cast(krow.ctypes.data + 8 * ka * pd, c_i64p), C.c_int64(at))
        at += cnt
    assert at == nnz
    krow[ntot] = nnz
    assert np.all(kcol >= 0) and not np.any(np.isnan(kval))               # every slot of K was written exactly
    return sp.csr_matrix((kval, kcol, krow), shape=(ntot, ntot))


def _setup(p, nels, knots=None, seed=0):
    from tigar_amd import tensorptap as TP
    kvs = knots if knots is not None else [O.uniform_knots(p, 0., 1., n) for n in nels]
    s = O.BSpline([p] * 3, kvs)
    Mo = O.generate_M_tensor(s)
    M1 = [O.generate_M_tensor(O.BSpline([p], [kvs[k]])).tocsr() for k in range(3)]
    wls = []
    for k in range(3):
        wl = TP.local_weights(M1[k], p, nels[k])
        assert wl is not None and TP.band_pattern_ok(M1[k], p, nels[k])
        wls.append(np.ascontiguousarray(wl))
    # FE matrix: random non-symmetric values on the element-coupling pattern of the Q_p grid
    pats = []
    for k in range(3):
        nfe = p * nels[k] + 1
        P1 = sp.lil_matrix((nfe, nfe))
        for e in range(nels[k]):
            P1[p * e:p * e + p + 1, p * e:p * e + p + 1] = 1.0
        pats.append(P1.tocsr())
    A = O.kron_dir0_fastest(pats).tocsr()
    A.sort_indices()
    rng = np.random.default_rng(seed)
    A.data = rng.standard_normal(A.nnz)
    zd = []
    for direction in range(3):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    return s, Mo, wls, A, zd


def _check(K, Ko):
    Ko = Ko.tocsr()
    Ko.sort_indices()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)      # pattern identical
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-13 * np.max(np.abs(Ko.data))


@pytest.mark.parametrize("p,nels", [(2, (3, 4, 2)), (3, (2, 3, 4)), (1, (3, 2, 4)), (2, (1, 1, 1)), (3, (1, 2, 1))])
def test_walks_reproduce_oracle_ptap(emu, p, nels):
    s, Mo, wls, A, zd = _setup(p, nels)
    nfe2, ncp2 = p * nels[2] + 1, nels[2] + p
    Ko = O.extract_matrix(Mo, A, zd, diag=2.5)
    K = _emulated_ptap(emu, p, nels, wls, A, [(0, nfe2)], [(0, ncp2)], zd, 2.5)
    _check(K, Ko)
    # no boundary conditions
    K0 = _emulated_ptap(emu, p, nels, wls, A, [(0, nfe2)], [(0, ncp2)], None, 1.0)
    _check(K0, O.extract_matrix(Mo, A, None))


@pytest.mark.parametrize("p,nel2", [(2, 5), (3, 5), (1, 6)])
def test_sub_slab_seams_and_plane_pieces(emu, p, nel2):
    """the z pass on dof-plane sub-slabs reading B2 planes from several pieces (the ring cache of the streamed
    path): every split of the dof planes and of the FE planes gives the same K"""
    nels = (2, 2, nel2)
    s, Mo, wls, A, zd = _setup(p, nels, seed=3)
    nfe2, ncp2 = p * nel2 + 1, nel2 + p
    Ko = O.extract_matrix(Mo, A, zd, diag=1.0)
    K_ref = _emulated_ptap(emu, p, nels, wls, A, [(0, nfe2)], [(0, ncp2)], zd, 1.0)
    _check(K_ref, Ko)
    for cut in range(1, ncp2):
        for pcut in (1, nfe2 // 2, nfe2 - 1):
            K = _emulated_ptap(emu, p, nels, wls, A, [(0, pcut), (pcut, nfe2)], [(0, cut), (cut, ncp2)], zd, 1.0)
            assert np.array_equal(K.indices, K_ref.indices) and np.array_equal(K.data, K_ref.data)   # bit-identical
    K3 = _emulated_ptap(emu, p, nels, wls, A, [(0, 2), (2, 3), (3, nfe2)], [(0, 1), (1, 2), (2, ncp2 - 1), (ncp2 - 1, ncp2)],
                        zd, 1.0)
    assert np.array_equal(K3.data, K_ref.data)


def test_non_uniform_knot_spacing(emu):
    p, nels = 2, (3, 2, 3)
    rng = np.random.default_rng(5)
    knots = []
    for n in nels:
        br = np.concatenate([[0.0], np.sort(rng.random(n - 1)), [1.0]])
        knots.append([0.0] * p + list(br) + [1.0] * p)
    s, Mo, wls, A, zd = _setup(p, nels, knots=knots, seed=7)
    K = _emulated_ptap(emu, p, nels, wls, A, [(0, p * nels[2] + 1)], [(0, nels[2] + p)], zd, 1.0)
    _check(K, O.extract_matrix(Mo, A, zd))


def test_other_patterns_are_declined(emu):
    """an FE matrix with one entry missing, one column moved, or a shorter row is not taken (flag -> general path)"""
    p, nels = 2, (2, 2, 2)
    s, Mo, wls, A, zd = _setup(p, nels)
    nfe2, ncp2 = p * nels[2] + 1, nels[2] + p
    A1 = A.copy().tolil()
    r = 37
    c = A1.rows[r][3]
    A1[r, c] = 0.0
    A1 = A1.tocsr()
    A1.eliminate_zeros()
    assert A1.nnz == A.nnz - 1
    assert _emulated_ptap(emu, p, nels, wls, A1, [(0, nfe2)], [(0, ncp2)], zd, 1.0) is None
    A2 = A.copy()
    A2.indices = A2.indices.copy()
    q = A2.indptr[50] + 2
    A2.indices[q] += 1 if A2.indices[q] + 1 not in A2.indices[A2.indptr[50]:A2.indptr[51]] else 0
    if A2.indices[q] != A.indices[q]:
        assert _emulated_ptap(emu, p, nels, wls, A2, [(0, nfe2)], [(0, ncp2)], zd, 1.0) is None


def test_structure_checks_reject_other_knot_vectors():
    from tigar_amd import tensorptap as TP
    p, nel = 2, 4
    # repeated interior knot (C^0 line): the dof numbering no longer advances by one per element
    kv = [0, 0, 0, 0.25, 0.5, 0.5, 0.75, 1, 1, 1]
    M1 = O.generate_M_tensor(O.BSpline([p], [kv])).tocsr()
    assert TP.local_weights(M1, p, nel) is None
    # periodic knot vector
    kvp = O.uniform_knots(p, 0., 1., nel, periodic=True)
    sp1 = O.BSpline([p], [kvp])
    M1p = O.generate_M_tensor(sp1).tocsr()
    assert TP.local_weights(M1p, p, nel) is None
    # the plain open vector passes
    M1o = O.generate_M_tensor(O.BSpline([p], [O.uniform_knots(p, 0., 1., nel)])).tocsr()
    assert TP.local_weights(M1o, p, nel) is not None and TP.band_pattern_ok(M1o, p, nel)


# ---------------------------------------------------------------------------------------------------------
# 2-D patches with nF fields on one basis (cfg4: biharmonic p = 4; cfg5: three fields): x pass with the field index
# as a dense third direction, then the final pass along direction 1 that writes K in CSR order
def _emulated_ptap_2d(emu, p, nels, nF, wls, A, zero_dofs, diag, ech=(0, 0)):
    """mirror of tg_tensor2_ptap (csrc/tg_ptap_tensor.hip) on host arrays"""
    W = 2 * p + 1
    nfe = [p * n + 1 for n in nels]
    ncp = [n + p for n in nels]
    tabs = [_tables(p, n) for n in nels]
    A = sp.csr_matrix(A)
    A.sort_indices()
    rowptr, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)
    t1 = int(tabs[1][0][nfe[1]])
    plane_b1 = W * nF * ncp[0] * t1
    b1 = np.full(plane_b1 * nF, np.nan)
    pb1 = (np.arange(nF + 1) * plane_b1).astype(np.int64)
    planes = np.arange(nF, dtype=np.int32)
    rps2 = (np.arange(nF + 1) * nF).astype(np.int32)
    first = True
    for n1 in (W, p + 1):
        lines = np.array([a for a in range(nfe[1]) if _rn(p, a, nfe[1]) == n1], dtype=np.int32)
        if not len(lines):
            continue
        L = max(1, 64 // (n1 * nF))
        bad = emu.emu_x2(p, _p(rowptr, c_i64p), _p(col, c_i32p), _p(val, c_f64p), nels[0], nfe[1], nF, _p(wls[0], c_f64p),
                         _p(tabs[0][0], c_i32p), _p(tabs[1][0], c_i32p), _p(rps2, c_i32p), _p(lines, c_i32p), len(lines), L,
                         n1, _p(planes, c_i32p), _p(b1, c_f64p), _p(pb1, c_i64p), ech[0], 1 if first else 0)
        first = False
        if bad:
            return None
    assert not np.any(np.isnan(b1))                                        # every block entry written (once per piece)
    ntot = nF * ncp[0] * ncp[1]
    nnz = nF * nF * int(tabs[0][1][ncp[0]]) * int(tabs[1][1][ncp[1]])
    mask = None
    if zero_dofs is not None and len(zero_dofs):
        mask = np.zeros(ntot, dtype=np.uint8)
        mask[np.asarray(zero_dofs)] = 1
    kcol = np.full(nnz, -1, dtype=np.int32)
    kval = np.full(nnz, np.nan)
    kdiag = np.full(ntot, np.nan)
    krow = np.zeros(ntot + 1, dtype=np.int64)
    emu.emu_y2(p, _p(b1, c_f64p), C.c_int64(plane_b1), nels[1], _p(wls[1], c_f64p), _p(tabs[1][0], c_i32p),
               _p(tabs[1][1], c_i32p), ncp[0], nF, _p(tabs[0][1], c_i32p), max(1, 64 // (W * nF)), ech[1],
               _p(kcol, c_i32p), _p(kval, c_f64p), _p(kdiag, c_f64p),
               _p(mask, C.POINTER(C.c_uint8)) if mask is not None else None, C.c_double(diag), _p(krow, c_i64p))
    krow[ntot] = nnz
    assert np.all(kcol >= 0) and not np.any(np.isnan(kval))
    K = sp.csr_matrix((kval, kcol, krow), shape=(ntot, ntot))
    assert np.array_equal(kdiag, K.diagonal())                             # the diagonal recorded on the way
    return K


def _setup_2d(p, nels, nF, seed=0):
    from tigar_amd import tensorptap as TP
    kvs = [O.uniform_knots(p, -1., 1., n) for n in nels]
    s = O.BSpline([p] * 2, kvs)
    Mo = O.generate_M_tensor(s, nfields=nF)
    wls = []
    for k in range(2):
        M1 = O.generate_M_tensor(O.BSpline([p], [kvs[k]])).tocsr()
        wl = TP.local_weights(M1, p, nels[k])
        assert wl is not None and TP.band_pattern_ok(M1, p, nels[k])
        wls.append(np.ascontiguousarray(wl))
    pats = []
    for k in range(2):
        nfe = p * nels[k] + 1
        P1 = sp.lil_matrix((nfe, nfe))
        for e in range(nels[k]):
            P1[p * e:p * e + p + 1, p * e:p * e + p + 1] = 1.0
        pats.append(P1.tocsr())
    pat = O.kron_dir0_fastest(pats).tocsr()
    A = sp.bmat([[pat] * nF for _ in range(nF)], format="csr")
    A.sort_indices()
    rng = np.random.default_rng(seed)
    A.data = rng.standard_normal(A.nnz)
    ncp = s.getNcp()
    zd = []
    for f in range(nF):
        for direction in range(2):
            for side in (0, 1):
                zd += [f * ncp + v for v in s.getSideDofs(direction, side, 2 if p > 2 else 1)]
    return s, Mo, wls, A, zd


@pytest.mark.parametrize("p,nels,nF", [(4, (3, 2), 1), (3, (4, 3), 3), (2, (5, 4), 2), (1, (3, 3), 1), (4, (1, 1), 1),
                                        (3, (2, 5), 1)])
def test_2d_walks_reproduce_oracle_ptap(emu, p, nels, nF):
    s, Mo, wls, A, zd = _setup_2d(p, nels, nF)
    Ko = O.extract_matrix(Mo, A, zd, diag=2.5)
    K = _emulated_ptap_2d(emu, p, nels, nF, wls, A, zd, 2.5)
    _check(K, Ko)
    _check(_emulated_ptap_2d(emu, p, nels, nF, wls, A, None, 1.0), O.extract_matrix(Mo, A, None))


@pytest.mark.parametrize("p,nF", [(4, 1), (3, 2), (2, 1)])
def test_2d_walks_in_pieces_are_bit_identical(emu, p, nF):
    """the walks cut into pieces of a few elements (each piece re-reads p elements to warm its ring up): same K bit for
    bit, every entry of B1 and K written exactly once"""
    nels = (9, 7)
    s, Mo, wls, A, zd = _setup_2d(p, nels, nF, seed=4)
    K_ref = _emulated_ptap_2d(emu, p, nels, nF, wls, A, zd, 1.0)
    _check(K_ref, O.extract_matrix(Mo, A, zd))
    for ech in ((1, 1), (2, 3), (4, 2), (8, 6), (5, 0), (0, 4)):
        K = _emulated_ptap_2d(emu, p, nels, nF, wls, A, zd, 1.0, ech=ech)
        assert np.array_equal(K.indices, K_ref.indices) and np.array_equal(K.data, K_ref.data)


def test_2d_other_patterns_are_declined(emu):
    p, nels, nF = 3, (2, 2), 2
    s, Mo, wls, A, zd = _setup_2d(p, nels, nF)
    A1 = A.copy().tolil()
    A1[20, A1.rows[20][2]] = 0.0
    A1 = A1.tocsr()
    A1.eliminate_zeros()
    assert _emulated_ptap_2d(emu, p, nels, nF, wls, A1, zd, 1.0) is None
    A2 = A.copy()
    A2.indices = A2.indices.copy()
    q = A2.indptr[31]                                                          # first entry of row 31
    assert A2.indices[q] >= 1
    A2.indices[q] -= 1                                                         # same row length, one column moved
    assert _emulated_ptap_2d(emu, p, nels, nF, wls, A2, zd, 1.0) is None
