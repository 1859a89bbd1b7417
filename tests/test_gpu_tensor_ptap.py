"""-m gpu: the tensor-pattern extractMatrix path (csrc/tg_tensor_body.h: line walks without column decode, LDS or
atomics) on the device, against the oracle's M^T A M + MatZeroRowsColumns, against the general kernels, and for
run-to-run bit-reproducibility of K (SURVEY.md section 7 hard part 4)."""
import os
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu


def _patch(p, nels, knots=None):
    import tigar_amd as t
    from tigar_amd import BSplines as B
    kvs = knots if knots is not None else [B.uniformKnots(p, 0., 1., n) for n in nels]
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kvs))
    sp0 = gen.getScalarSpline(0)
    for direction in range(3):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    return gen, t.ExtractedSpline(gen, 2 * p)


def _random_fe_matrix(p, nels, seed=0):
    pats = []
    for k in range(3):
        nfe = p * nels[k] + 1
        P1 = sp.lil_matrix((nfe, nfe))
        for e in range(nels[k]):
            P1[p * e:p * e + p + 1, p * e:p * e + p + 1] = 1.0
        pats.append(P1.tocsr())
    A = O.kron_dir0_fastest(pats).tocsr()
    A.sort_indices()
    A.data = np.random.default_rng(seed).standard_normal(A.nnz)
    return A


@pytest.mark.parametrize("p,nels", [(2, (3, 4, 2)), (3, (2, 3, 4)), (1, (3, 2, 4)), (2, (1, 1, 1)), (3, (5, 4, 6)), (2, (9, 7, 8))])
def test_device_walks_vs_oracle(p, nels):
    from tigar_amd.tensorptap import TensorPtAP
    from tigar_amd import device as dev
    gen, spline = _patch(p, nels)
    plan = TensorPtAP.for_extraction(spline._kron)
    assert plan is not None                                   # the patch qualifies
    A = _random_fe_matrix(p, nels, seed=p)
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., n) for n in nels])
    Mo = O.generate_M_tensor(s)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd, diag=2.5)
    Ad = dev.DeviceCSR.from_scipy(A)
    nfe2, ncp2 = p * nels[2] + 1, nels[2] + p
    piece = plan.planes(Ad, 0, 0, nfe2)
    assert piece is not None
    K = plan.zstage([piece], 0, ncp2, zd, 2.5).to_scipy()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-13 * np.max(np.abs(Ko.data))
    # through the public API (extractMatrix picks the same path), three times: bit-identical
    K1 = spline.extractMatrix(A, diag=2.5).to_scipy()
    assert np.array_equal(K1.data, K.data) and np.array_equal(K1.indices, K.indices)
    for _ in range(2):
        assert np.array_equal(spline.extractMatrix(A, diag=2.5).to_scipy().data, K.data)
    # sub-slabs of dof planes reading planes from two pieces: bit-identical to the single pass
    if nfe2 >= 3 and ncp2 >= 2:
        cut, pcut = ncp2 // 2, nfe2 // 2
        pa, pb = plan.planes(Ad, 0, 0, pcut), plan.planes(Ad, 0, pcut, nfe2)
        Ka = plan.zstage([pa, pb], 0, cut, zd, 2.5).to_scipy()
        Kb = plan.zstage([pa, pb], cut, ncp2, zd, 2.5).to_scipy()
        assert np.array_equal(sp.vstack([Ka, Kb]).tocsr().data, K.data)
    # the general kernels agree to rounding
    os.environ["TIGAR_PTAP_TENSOR"] = "0"
    try:
        gen2, spline2 = _patch(p, nels)
        Kg = spline2.extractMatrix(A, diag=2.5).to_scipy()
    finally:
        os.environ.pop("TIGAR_PTAP_TENSOR", None)
    assert np.array_equal(Kg.indices, K.indices)
    assert np.max(np.abs(Kg.data - K.data)) <= 1e-12 * np.max(np.abs(K.data))


def test_other_patterns_fall_back_to_the_general_kernels():
    """an FE matrix with an extra coupling (contact term, demos/kl-shell-svk/reef-knot.py:460-467) or a missing
    entry is not taken by the fast path; extractMatrix still returns M^T A M"""
    p, nels = 2, (3, 3, 3)
    gen, spline = _patch(p, nels)
    A = _random_fe_matrix(p, nels, seed=11).tolil()
    A[5, A.shape[0] - 3] = 0.75                                  # couples two far-apart nodes
    A = A.tocsr()
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., n) for n in nels])
    Mo = O.generate_M_tensor(s)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd)
    K = spline.extractMatrix(A).to_scipy()
    assert np.array_equal(K.indices, Ko.indices)
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-12 * np.max(np.abs(Ko.data))
    from tigar_amd.tensorptap import TensorPtAP
    from tigar_amd import device as dev
    plan = TensorPtAP.for_extraction(spline._kron)
    assert plan.planes(dev.DeviceCSR.from_scipy(A), 0, 0, p * nels[2] + 1) is None


def test_streamed_sub_slabs_with_tensor_ring_match_resident():
    """SlabHotPath with the tensor ring (B2 planes kept across sub-slabs, rows of K written straight into the
    builder at closed-form positions) against the resident product, and a declined pattern in the streamed path"""
    from tigar_amd import forms as F, device as dev
    from tigar_amd.dist import SlabHotPath
    p, nels = 3, (6, 5, 9)
    gen, spline = _patch(p, nels)
    lap = F.LaplaceForm()
    A = lap.assemble_matrix(gen.V)
    K_res = spline.extractMatrix(A).to_scipy()
    zd = list(spline.zeroDofs)
    basis = gen.getScalarSpline(0)
    for sub in (1, 2, 5):
        path = SlabHotPath(basis, gen.V.grids[0], sub_planes=sub)
        K = path.assemble_matrix(lambda a, b: lap.assemble_matrix(gen.V, a, b), zd, 1.0).to_scipy()
        assert np.array_equal(K.indptr, K_res.indptr) and np.array_equal(K.indices, K_res.indices)
        assert np.array_equal(K.data, K_res.data)                   # same walks, same order: bit-identical
    # a producer whose rows are not on the pattern: the streamed assembly restarts on the general stages
    As = A.to_scipy().tolil()
    As[7, As.shape[0] - 5] = 0.5
    As = As.tocsr()

    def rows(a, b):
        return dev.DeviceCSR.from_scipy(As[a:b])
    path = SlabHotPath(basis, gen.V.grids[0], sub_planes=3)
    K = path.assemble_matrix(rows, zd, 1.0).to_scipy()
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., n) for n in nels])
    Ko = O.extract_matrix(O.generate_M_tensor(s), As, zd)
    assert np.array_equal(K.indices, Ko.indices)
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-12 * np.max(np.abs(Ko.data))


@pytest.mark.parametrize("d,p,nel,factored", [(3, 3, 12, "1"), (2, 5, 40, "1"), (2, 3, 120, "1"), (3, 2, 10, "0")])
def test_general_kernels_are_bit_reproducible(monkeypatch, d, p, nel, factored):
    """The general PtAP kernels (other patterns, p > 4, arbitrary M; TIGAR_PTAP_TENSOR=0 selects them) give the same K to
    the last bit in every run.  The line kernels (3-D, one direction at a time) own their accumulators wave by wave; the
    box kernel (2-D: several waves scatter into one LDS box; up to round 2 two thirds of the entries of these 2-D cases
    changed their last bits from run to run) and the hash kernel (TIGAR_PTAP_FACTORED=0: nothing assumed about M) add
    integers on a grid derived from a bound of the row's accumulators (tg_fix, csrc/tg_common.h).  K stays within
    rounding of the tensor-pattern path where that applies."""
    import scipy.sparse as sp
    import tigar_amd as t
    from tigar_amd import BSplines as B
    monkeypatch.setenv("TIGAR_PTAP_TENSOR", "0")
    monkeypatch.setenv("TIGAR_PTAP_FACTORED", factored)
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, [B.uniformKnots(p, 0., 1., nel)] * d))
    spline = t.ExtractedSpline(gen, 2 * p)
    i = np.arange(p * nel + 1)
    C1 = sp.csr_matrix(np.abs(i[:, None] - i[None, :]) <= p)           # a band that contains the element coupling
    P = C1
    for _ in range(d - 1):
        P = sp.kron(C1, P, format="csr")
    A = P.astype(np.float64).tocsr()
    A.data = np.random.default_rng(3).standard_normal(A.nnz)
    runs = [spline.extractMatrix(A).to_scipy() for _ in range(4)]
    for K in runs[1:]:
        assert np.array_equal(K.indptr, runs[0].indptr) and np.array_equal(K.indices, runs[0].indices)
        assert np.array_equal(K.data.view(np.int64), runs[0].data.view(np.int64))
    M = gen.M.to_scipy()
    Ko = (M.T @ A @ M).tocsr()
    assert abs(runs[0] - Ko).max() <= 1e-13 * abs(Ko).max()


@pytest.mark.parametrize("p,nels", [(3, (4, 3, 5)), (2, (6, 5, 4))])
def test_pattern_certificate_of_library_assembled_matrices(p, nels):
    """An FE matrix assembled by this library's own kernel (forms.LaplaceForm -> tg_kron_sum_csr) carries a certificate
    of its sparsity pattern; the x pass then does not read the column indices again.  K is bit-identical with and
    without the verification; the same matrix uploaded from the host has no certificate and is verified; a certified
    matrix of ANOTHER node grid is not mistaken for this one's."""
    from tigar_amd.tensorptap import TensorPtAP
    from tigar_amd import device as dev, forms as F
    certified = lambda: dev.prof_get(3)[1]
    gen, spline = _patch(p, nels)
    plan = TensorPtAP.for_extraction(spline._kron)
    nfe2, ncp2 = p * nels[2] + 1, nels[2] + p
    A = F.LaplaceForm().assemble_matrix(spline.V)
    zd = list(spline.zeroDofs)
    n0 = certified()
    K_cert = plan.zstage([plan.planes(A, 0, 0, nfe2)], 0, ncp2, zd, 1.0).to_scipy()
    assert certified() == n0 + 1
    os.environ["TIGAR_PTAP_VERIFY"] = "1"
    try:
        K_ver = plan.zstage([plan.planes(A, 0, 0, nfe2)], 0, ncp2, zd, 1.0).to_scipy()
    finally:
        del os.environ["TIGAR_PTAP_VERIFY"]
    assert certified() == n0 + 1
    assert np.array_equal(K_cert.indices, K_ver.indices) and np.array_equal(K_cert.data, K_ver.data)
    # row blocks: the certificate names the first row of the block
    pf = (p * nels[0] + 1) * (p * nels[1] + 1)
    z0 = p                                       # (a whole element below)
    Ab = F.LaplaceForm().assemble_matrix(spline.V, z0 * pf, nfe2 * pf)
    assert plan.planes(Ab, z0 * pf, z0, nfe2) is not None and certified() == n0 + 2
    # no certificate on a matrix that came through the host
    Ah = dev.DeviceCSR.from_scipy(A.to_scipy())
    K_h = plan.zstage([plan.planes(Ah, 0, 0, nfe2)], 0, ncp2, zd, 1.0).to_scipy()
    assert certified() == n0 + 2 and np.array_equal(K_h.data, K_cert.data)
    # a certified matrix of another grid with the same number of nodes per plane and planes: other 1-D patterns
    if p == 2:
        gen3, spline3 = _patch(1, tuple(2 * n for n in nels))          # degree 1 on twice the elements: same node counts
        A3 = F.LaplaceForm().assemble_matrix(spline3.V)
        assert A3.shape == A.shape
        assert plan.planes(A3, 0, 0, nfe2) is None                     # verified, found different, declined
        assert certified() == n0 + 2


@pytest.mark.parametrize("p,nels,nF", [(2, (3, 4, 2), 3), (3, (3, 2, 3), 2), (1, (4, 3, 5), 3)])
def test_several_fields_on_one_basis_go_block_by_block(p, nels, nF):
    """EqualOrderSpline(nFields > 1) in 3-D (elasticity-like): M = diag(M_s, ..., M_s) with dofs numbered field after
    field, so block (i, j) of M^T A M is M_s^T A_ij M_s -- the scalar tensor-pattern passes on the blocks cut out of A
    (tg_csr_block), put together (tg_csr_from_blocks), MatZeroRowsColumns on the whole.  Against the oracle's product
    with the block-diagonal M (pattern and values), with a pair of uncoupled fields (an empty block), a non-symmetric A,
    and against the general kernels."""
    import tigar_amd as t
    from tigar_amd import BSplines as B, device as dev
    kvs = [B.uniformKnots(p, 0., 1., n) for n in nels]
    gen = t.EqualOrderSpline(nF, B.ExplicitBSplineControlMesh([p] * 3, kvs))
    sp0 = gen.getScalarSpline(0)
    for f in range(nF):
        gen.addZeroDofs(f, sp0.getSideDofs(f % 3, 0))
        gen.addZeroDofs(f, sp0.getSideDofs((f + 1) % 3, 1, nLayers=2))
    spline = t.ExtractedSpline(gen, 2 * p)
    assert spline._kron is None and spline._kron_scalar is not None
    rng = np.random.default_rng(7 * p + nF)
    rows = []
    for i in range(nF):
        row = []
        for j in range(nF):
            blk = _random_fe_matrix(p, nels, seed=100 * i + j)
            if nF == 3 and (i, j) in ((0, 2), (2, 0)):
                blk = sp.csr_matrix(blk.shape)                        # fields 0 and 2 not coupled
            row.append(blk)
        rows.append(row)
    A = sp.bmat(rows, format="csr")
    A.sort_indices()
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., n) for n in nels])
    Mo = O.generate_M_tensor(s, nfields=nF)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd, diag=3.0)
    calls = []
    orig = spline._extract_matrix_by_field_blocks
    spline._extract_matrix_by_field_blocks = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    K = spline.extractMatrix(A, diag=3.0).to_scipy()
    assert calls == [1]
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-13 * np.max(np.abs(Ko.data))
    assert np.array_equal(spline.extractMatrix(A, diag=3.0).to_scipy().data, K.data)      # bit-reproducible
    os.environ["TIGAR_PTAP_FACTORED"] = "0"
    try:
        Kg = spline.extractMatrix(A, diag=3.0).to_scipy()
    finally:
        del os.environ["TIGAR_PTAP_FACTORED"]
    assert np.array_equal(Kg.indptr, K.indptr) and np.array_equal(Kg.indices, K.indices)
    assert np.max(np.abs(Kg.data - K.data)) <= 1e-12 * np.max(np.abs(K.data))
    # a block without the element-coupling pattern is multiplied by the general kernels -- that block alone, on the
    # scalar operands; also with the tensor passes switched off altogether
    Ar = A.tolil()
    Ar[1, A.shape[1] - 2] = 0.5
    Ar = Ar.tocsr()
    Ar.sort_indices()
    Kro = O.extract_matrix(Mo, Ar, zd, diag=3.0)
    Kr = spline.extractMatrix(Ar, diag=3.0).to_scipy()
    assert calls == [1, 1, 1]
    assert np.array_equal(Kr.indptr, Kro.indptr) and np.array_equal(Kr.indices, Kro.indices)
    assert np.max(np.abs(Kr.data - Kro.data)) <= 1e-12 * np.max(np.abs(Kro.data))
    Kb = orig(dev.DeviceCSR.from_scipy(A), np.asarray(zd), 3.0, False).to_scipy()
    assert np.array_equal(Kb.indices, K.indices) and np.max(np.abs(Kb.data - K.data)) <= 1e-12 * np.max(np.abs(K.data))


@pytest.mark.parametrize("p,nels", [(2, (4, 3, 5)), (3, (3, 3, 2)), (1, (5, 4, 3))])
def test_matrix_with_entries_outside_the_pattern_is_split(p, nels):
    """An FE matrix with couplings added by hand (demos/kl-shell-svk/reef-knot.py:466 adds contact terms) and with some
    entries of the element-coupling pattern missing: the part on the pattern goes through the line walks, the rest
    through the general kernels, the products are added on the union of their patterns and the boundary conditions
    applied to the sum.  Against the oracle's M^T A M (values; pattern = structural pattern of the product), and the
    pieces themselves against A.  The split itself pads the pattern part with stored zeros where A has no entry; since
    round 4 only a matrix that stores the WHOLE pattern takes it (plus whatever was added), one with entries missing takes the
    general stages, so that the product's pattern is always the structural product of what A stores."""
    from tigar_amd.tensorptap import TensorPtAP
    from tigar_amd import device as dev
    gen, spline = _patch(p, nels)
    plan = TensorPtAP.for_extraction(spline._kron)
    A0 = _random_fe_matrix(p, nels, seed=11)
    n = A0.shape[0]
    rng = np.random.default_rng(4)
    # drop a few entries of the pattern, add far couplings (some rows get many), keep it non-symmetric
    A = A0.tolil()
    for r in rng.choice(n, size=12, replace=False):
        cols = A0.indices[A0.indptr[r]:A0.indptr[r + 1]]
        cols = cols[cols != r]                      # (a missing DIAGONAL entry is another matter: see the docstring)
        A[r, cols[rng.integers(0, cols.size)]] = 0.0
    extra_rows = rng.choice(n, size=15, replace=False)
    for r in extra_rows:
        for c in rng.choice(n, size=int(rng.integers(1, 40)), replace=False):
            A[r, c] = rng.standard_normal()
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    Ad = dev.DeviceCSR.from_scipy(A)
    assert plan.planes(Ad, 0, 0, p * nels[2] + 1) is None                    # not the pattern
    on, off = plan.split(Ad)
    on_s, off_s = on.to_scipy(), off.to_scipy()
    assert on_s.nnz == A0.nnz and np.array_equal(on_s.indices, A0.indices)    # exactly the pattern, zeros where A has none
    assert abs(on_s + off_s - A).max() == 0.0 and off_s.nnz > 0
    assert off_s.multiply(abs(A0) > 0).nnz == 0                               # nothing of the remainder lies on the pattern
    s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., m) for m in nels])
    Mo = O.generate_M_tensor(s)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd, diag=2.0)
    K = spline.extractMatrix(A, diag=2.0).to_scipy()
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # pattern: the structural product of what A stores (round 4: a matrix that LACKS entries of the pattern takes the general
    # stages -- the split would pad its on-part with stored zeros and K would get entries MatPtAP does not create)
    K.sort_indices(), Ko.sort_indices()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    # with nothing missing the split is taken (the line walks run) and the pattern is again the structural product
    Af = (A0 + off_s).tocsr()
    Af.sort_indices()
    dev.prof_reset()
    Kf = spline.extractMatrix(Af, diag=2.0).to_scipy()
    assert dev.prof_get(5)[1] > 0
    Kfo = O.extract_matrix(Mo, Af, zd, diag=2.0)
    Kf.sort_indices(), Kfo.sort_indices()
    assert abs(Kf - Kfo).max() <= 1e-12 * abs(Kfo).max()
    assert np.array_equal(Kf.indptr, Kfo.indptr) and np.array_equal(Kf.indices, Kfo.indices)
    # the same result with the split switched off (general line kernels on the whole matrix)
    os.environ["TIGAR_PTAP_SPLIT"] = "0"
    try:
        Kg = spline.extractMatrix(A, diag=2.0).to_scipy()
    finally:
        del os.environ["TIGAR_PTAP_SPLIT"]
    assert abs(Kg - K).max() <= 1e-12 * abs(K).max()
    # A + B on the union pattern (tg_csr_add) on its own
    B1, B2 = _rand_pair(rng, 300, 200)
    S = dev.DeviceCSR.from_scipy(B1).add(dev.DeviceCSR.from_scipy(B2)).to_scipy()
    R = (B1 + B2).tocsr()
    assert abs(S - R).max() < 1e-15 and S.has_sorted_indices
    U = ((abs(B1) > 0).astype(int) + (abs(B2) > 0).astype(int)).tocsr()
    assert S.nnz == U.nnz


def _rand_pair(rng, n, m):
    a = sp.random(n, m, density=0.05, random_state=np.random.RandomState(1), format="csr")
    b = sp.random(n, m, density=0.01, random_state=np.random.RandomState(2), format="csr")
    a.sort_indices(), b.sort_indices()
    return a, b


# ---------------------------------------------------------------------------------------------------------
# 2-D patches (one or several fields on one basis; degrees up to 4): tg_tensor2_ptap
def _patch2(p, nels, nF, lo=-1.0):
    import tigar_amd as t
    from tigar_amd import BSplines as B
    kvs = [B.uniformKnots(p, lo, 1., n) for n in nels]
    gen = t.EqualOrderSpline(nF, B.ExplicitBSplineControlMesh([p] * 2, kvs))
    for f in range(nF):
        sp0 = gen.getScalarSpline(f)
        for direction in range(2):
            for side in (0, 1):
                gen.addZeroDofs(f, sp0.getSideDofs(direction, side, nLayers=2 if p > 2 else 1))
    return gen, t.ExtractedSpline(gen, 2 * p)


def _random_fe_matrix2(p, nels, nF, seed=0):
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
    A.data = np.random.default_rng(seed).standard_normal(A.nnz)
    return A


@pytest.mark.parametrize("p,nels,nF", [(4, (3, 2), 1), (4, (40, 37), 1), (3, (4, 3), 3), (3, (33, 21), 3), (2, (5, 4), 2),
                                        (1, (3, 3), 1), (4, (1, 1), 1), (2, (70, 3), 4)])
def test_2d_device_walks_vs_oracle(p, nels, nF):
    from tigar_amd.tensorptap import TensorPtAP2D
    from tigar_amd import device as dev
    gen, spline = _patch2(p, nels, nF)
    kx = spline._kron if nF == 1 else spline._kron_scalar
    plan = TensorPtAP2D.for_extraction(kx, nF)
    assert plan is not None                                   # the patch qualifies
    A = _random_fe_matrix2(p, nels, nF, seed=p)
    s = O.BSpline([p] * 2, [O.uniform_knots(p, -1., 1., n) for n in nels])
    Mo = O.generate_M_tensor(s, nfields=nF)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd, diag=2.5)
    Ad = dev.DeviceCSR.from_scipy(A)
    Kd = plan.ptap(Ad, zd, 2.5)
    assert Kd is not None
    K = Kd.to_scipy()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert np.max(np.abs(K.data - Ko.data)) <= 1e-13 * np.max(np.abs(Ko.data))
    # through the public API (extractMatrix picks the same path), three times: bit-identical
    dev.prof_reset()
    for _ in range(3):
        K1 = spline.extractMatrix(A, diag=2.5).to_scipy()
        assert np.array_equal(K1.data, K.data) and np.array_equal(K1.indices, K.indices)
    # no boundary conditions
    K0 = spline.extractMatrix(A, applyBCs=False).to_scipy()
    Ko0 = O.extract_matrix(Mo, A, None)
    assert np.array_equal(K0.indices, Ko0.indices) and np.max(np.abs(K0.data - Ko0.data)) <= 1e-13 * np.max(np.abs(Ko0.data))
    # the general kernels agree to rounding; the Jacobi solve uses the diagonal recorded by the final pass
    os.environ["TIGAR_PTAP_TENSOR"] = "0"
    try:
        gen2, spline2 = _patch2(p, nels, nF)
        Kg = spline2.extractMatrix(A, diag=2.5).to_scipy()
    finally:
        os.environ.pop("TIGAR_PTAP_TENSOR", None)
    assert np.array_equal(Kg.indices, K.indices)
    assert np.max(np.abs(Kg.data - K.data)) <= 1e-12 * np.max(np.abs(K.data))


def test_2d_other_patterns_fall_back_to_the_general_kernels():
    """a block with an entry outside the element-coupling pattern (or a missing one): the fast path declines on the
    device, extractMatrix still returns the oracle's product through the general kernels"""
    p, nels, nF = 3, (6, 5), 2
    gen, spline = _patch2(p, nels, nF)
    A = _random_fe_matrix2(p, nels, nF, seed=2).tolil()
    A[3, A.shape[1] - 2] = 0.7                                 # a coupling added by hand
    A = A.tocsr()
    s = O.BSpline([p] * 2, [O.uniform_knots(p, -1., 1., n) for n in nels])
    Mo = O.generate_M_tensor(s, nfields=nF)
    zd = list(spline.zeroDofs)
    Ko = O.extract_matrix(Mo, A, zd)
    K = spline.extractMatrix(A).to_scipy()
    K.sort_indices()
    D = (K - Ko).tocsr()
    assert abs(D).max() <= 1e-12 * abs(Ko).max()
    # certified scalar matrix (assembled by the library on this grid): K equal with and without verification
    from tigar_amd import forms as F, device as dev
    gen1, spline1 = _patch2(4, (12, 9), 1)
    A1 = F.BiharmonicForm().assemble_matrix(spline1.V)
    dev.prof_reset()
    Ka = spline1.extractMatrix(A1).to_scipy()
    assert dev.prof_get(3)[1] == 1                              # pattern taken from the certificate
    os.environ["TIGAR_PTAP_VERIFY"] = "1"
    try:
        Kb = spline1.extractMatrix(A1).to_scipy()
    finally:
        os.environ.pop("TIGAR_PTAP_VERIFY", None)
    assert dev.prof_get(3)[1] == 1
    assert np.array_equal(Ka.data, Kb.data) and np.array_equal(Ka.indices, Kb.indices)


def test_kronecker_sum_forms_are_fused_into_the_x_pass_bit_for_bit():
    """assembleMatrix(form) with a form that is a Kronecker sum of 1-D matrices (LaplaceForm, MassForm on the identity
    geometry) on a streamed / implicit patch: the FE matrix is never written -- the x pass forms its entries exactly as
    tg_kron_sum_csr would have -- and K equals, bit for bit, the K of the unfused path (TIGAR_PTAP_FUSED=0: row blocks
    materialised, then read), over several sub-slabs, with boundary conditions, for one- and three-term forms."""
    import tigar_amd as t
    from tigar_amd import BSplines as B, forms as F
    # (the long x directions make both x passes walk in PIECES -- the fused one in pieces of ~12 KB of scalar tables: 21
    #  elements at p = 3 with three terms, 85 at p = 2 with one, 118 at p = 1; the other in two halves from 64 elements on --
    #  with a last piece shorter than the others)
    for p, nels, form in ((3, (6, 5, 9), F.LaplaceForm()), (2, (7, 8, 10), F.MassForm()), (1, (4, 4, 6), F.LaplaceForm()),
                          (3, (50, 3, 5), F.LaplaceForm()), (2, (95, 3, 4), F.MassForm()), (1, (130, 3, 4), F.LaplaceForm()),
                          (3, (70, 2, 4), F.MassForm())):
        Ks = []
        for fused in ("1", "0"):
            os.environ["TIGAR_PTAP_FUSED"] = fused
            os.environ["TIGAR_IMPLICIT_M"] = "1"
            os.environ["TIGAR_SUB_PLANES"] = "4"
            try:
                kvs = [B.uniformKnots(p, 0., 1., n) for n in nels]
                gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * 3, kvs))
                sp0 = gen.getScalarSpline(0)
                for direction in range(3):
                    for side in (0, 1):
                        gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
                spline = t.ExtractedSpline(gen, 2 * p)
                spline.stage_timers = {}
                K = spline.assembleMatrix(form, diag=2.0)
                assert getattr(gen.M, "is_implicit", False)
                # the fused path produces no FE input at all
                assert (spline.stage_timers.get("input", 0.0) < 1e-3) == (fused == "1") or fused == "0"
                Ks.append(K.to_scipy())
            finally:
                for k in ("TIGAR_PTAP_FUSED", "TIGAR_IMPLICIT_M", "TIGAR_SUB_PLANES"):
                    os.environ.pop(k, None)
        assert np.array_equal(Ks[0].indptr, Ks[1].indptr) and np.array_equal(Ks[0].indices, Ks[1].indices)
        assert np.array_equal(Ks[0].data, Ks[1].data)
        # and both are the oracle's product
        s = O.BSpline([p] * 3, [O.uniform_knots(p, 0., 1., n) for n in nels])
        Mo = O.generate_M_tensor(s)
        Ao = form.assemble_matrix(spline.V).to_scipy()
        Ko = O.extract_matrix(Mo, Ao, list(spline.zeroDofs), diag=2.0)
        assert np.array_equal(Ks[0].indices, Ko.indices)
        assert np.max(np.abs(Ks[0].data - Ko.data)) <= 1e-12 * np.max(np.abs(Ko.data))


# ---- round 5: blocks with different spline bases on the row and the column side (fields of compatible splines) -----------
def _compat_fields(kind, degs, nels):
    import tigar_amd as t
    from tigar_amd import BSplines as B, common as tc
    from tigar_amd.compatibleSplines import BSplineCompat
    from tigar_amd.kronptap import KronExtraction
    kv = [B.uniformKnots(degs[k], 0., 1. + 0.25 * k, nels[k]) for k in range(3)]
    gen = BSplineCompat(tc.selfcomm, B.ExplicitBSplineControlMesh(list(degs), kv), kind, list(degs))
    kxs = [KronExtraction(gen.getFieldSpline(f), gen.V.grids[f]) for f in range(3)]
    return gen, kxs


@pytest.mark.parametrize("kind,degs,nels", [("RT", (1, 1, 1), (3, 4, 5)), ("RT", (2, 2, 2), (3, 2, 4)), ("N", (1, 1, 1), (2, 3, 3)),
                                            ("RT", (1, 1, 1), (1, 1, 2))])
def test_tensor_walks_with_different_bases_on_rows_and_columns(kind, degs, nels):
    """K_fg = M_f^T A_fg M_g for the components of a compatible B-spline (tIGAr/compatibleSplines.py:21-66): all fields
    extract to one Q_P grid (P = base degree + 1), the bases differ in degree per direction.  The line walks with separate
    row- and column-side weights (padded to P + 1 functions per element), the last pass writing the true pattern: against
    scipy's product of the Kronecker operators, pattern = the oracle's structural product, streamed in pieces of planes."""
    import scipy.sparse as sps
    from tigar_amd import device as dev, forms as F
    from tigar_amd.tensorptap import TensorPtAP
    from oracle import tigar_oracle as O
    dev.device_info()
    gen, kxs = _compat_fields(kind, degs, nels)
    g = gen.V.grids[0]
    P = g.degree
    assert P == max(degs) + 1 and all(np.array_equal(a, b) for gi in gen.V.grids for a, b in zip(gi.axes, g.axes))
    # an FE matrix on the element-coupling pattern of the Q_P grid with arbitrary (non-symmetric) values
    V1 = type(gen.V)([g], gen.V.element)
    A = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
    rng = np.random.default_rng(5)
    A.data = A.data + 0.3 * rng.standard_normal(A.nnz)
    Ad = dev.DeviceCSR.from_scipy(A)
    nz = g.shape()[-1]
    Ms = [sps.kron(kx.M1[2], sps.kron(kx.M1[1], kx.M1[0])).tocsr() for kx in kxs]
    for Mk in Ms:
        Mk.eliminate_zeros()           # (scipy's kron of CSR operands stores whole blocks)
        Mk.sort_indices()
    dev.prof_reset()
    for f in range(3):
        for gg in range(3):
            plan = TensorPtAP.for_pair(kxs[f], kxs[gg])
            assert plan is not None
            cut = max(1, nz // 2)
            pieces = [plan.planes(Ad, 0, 0, cut), plan.planes(Ad, 0, cut, nz)] if cut < nz else [plan.planes(Ad, 0, 0, nz)]
            assert all(pc is not None for pc in pieces)
            ncr2 = kxs[f].ncp[2]
            K = plan.zstage(pieces, 0, ncr2).to_scipy()
            assert K.nnz == plan.k_nnz(0, ncr2)
            # rows of a range of dof planes only
            if ncr2 > 2:
                Kp = plan.zstage(pieces, 1, ncr2 - 1).to_scipy()
                pd = kxs[f].ncp[0] * kxs[f].ncp[1]
                assert abs(Kp - K[pd:(ncr2 - 1) * pd]).max() == 0.0
            # the oracle's product of block (f, g): pattern (structural product) and values
            Mf, Mg = Ms[f], Ms[gg]
            Kr = (Mf.T @ A @ Mg).tocsr()
            ones = lambda X: sps.csr_matrix((np.ones(X.nnz), X.indices, X.indptr), shape=X.shape)
            S = (ones(Mf).T @ ones(A) @ ones(Mg)).tocsr()
            S.sort_indices()
            assert K.shape == Kr.shape
            assert np.array_equal(K.indptr, S.indptr) and np.array_equal(K.indices, S.indices)
            assert abs(K - Kr).max() <= 1e-12 * abs(Kr).max()
    assert dev.prof_get(5)[1] >= 9


def _rt_problem(degs, nels, comm=None):
    """BSplineCompat RT space on an identity-geometry patch with the demos' normal-direction boundary conditions
    (demos/taylor-green/taylor-green-3d.py:42-50) and the oracle's block-diagonal extraction operator"""
    import scipy.sparse as sps
    import tigar_amd as t
    from tigar_amd import BSplines as B, common as tc
    from tigar_amd.compatibleSplines import BSplineCompat
    kv = [B.uniformKnots(degs[k], 0., 1. + 0.25 * k, nels[k]) for k in range(3)]
    gen = BSplineCompat(comm if comm is not None else tc.selfcomm, B.ExplicitBSplineControlMesh(list(degs), kv), "RT", list(degs))
    for field in range(3):
        sp_f = gen.getFieldSpline(field)
        for side in (0, 1):
            gen.addZeroDofs(field, sp_f.getSideDofs(field, side))
    blocks = []
    for i in range(3):
        f = gen.getFieldSpline(i)
        blocks.append(O.generate_M_tensor(O.BSpline([s1.p for s1 in f.splines], [np.asarray(s1.knots) for s1 in f.splines])))
    return gen, sps.block_diag(blocks, format="csr")


@pytest.mark.parametrize("degs,nels", [((1, 1, 1), (4, 3, 5)), ((2, 2, 2), (2, 3, 3))])
def test_extract_matrix_on_a_compatible_spline_takes_the_pair_walks(degs, nels, monkeypatch):
    """extractMatrix on BSplineCompat("RT") -- the space of the reference's only Krylov + MPI demos -- with an assembled
    block matrix (the linear-elasticity form on the common Q_P grid): every block through the line walks with different
    row / column bases, K against the oracle (pattern and values), then with a hand-added coupling off the pattern
    (general kernels for the blocks concerned)."""
    import tigar_amd as t
    from tigar_amd import device as dev, forms as F
    gen, Mo = _rt_problem(degs, nels)
    assert gen._kron_fields is not None and len(gen._kron_fields) == 3
    spline = t.ExtractedSpline(gen, 2 * (max(degs) + 1))
    A = F.ElasticityForm(1.3, 0.7).assemble_matrix(spline.V)
    zd = [int(i) for i in gen.zeroDofsArray()]
    dev.prof_reset()
    K = spline.extractMatrix(A, diag=2.5).to_scipy()
    assert dev.prof_get(5)[1] == 9                       # nine blocks, nine final passes of the walks
    Kr = O.extract_matrix(Mo, A.to_scipy(), zd, diag=2.5)
    assert np.array_equal(K.indptr, Kr.indptr) and np.array_equal(K.indices, Kr.indices)
    assert abs(K - Kr).max() <= 1e-12 * abs(Kr).max()
    # a coupling dolfin would not have assembled (hand-added): the blocks that carry it go through the general kernels
    Ah = A.to_scipy().tolil()
    Ah[3, Ah.shape[1] - 5] = 0.25
    Ah = Ah.tocsr()
    K2 = spline.extractMatrix(Ah, diag=2.5).to_scipy()
    K2r = O.extract_matrix(Mo, Ah, zd, diag=2.5)
    assert np.array_equal(K2.indptr, K2r.indptr) and np.array_equal(K2.indices, K2r.indices)
    assert abs(K2 - K2r).max() <= 1e-12 * abs(K2r).max()
    # the same product with the walks switched off
    monkeypatch.setenv("TIGAR_PTAP_TENSOR", "0")
    K3 = spline.extractMatrix(A, diag=2.5).to_scipy()
    assert np.array_equal(K3.indices, Kr.indices) and abs(K3 - Kr).max() <= 1e-12 * abs(Kr).max()


# ---- round 6: the same for 2-D compatible splines (demos/taylor-green/taylor-green-2d.py) ---------------------------------------
def _compat_fields_2d(kind, degs, nels):
    from tigar_amd import BSplines as B, common as tc
    from tigar_amd.compatibleSplines import BSplineCompat
    from tigar_amd.kronptap import KronExtraction
    kv = [B.uniformKnots(degs[k], 0., 1. + 0.25 * k, nels[k]) for k in range(2)]
    gen = BSplineCompat(tc.selfcomm, B.ExplicitBSplineControlMesh(list(degs), kv), kind, list(degs))
    kxs = [KronExtraction(gen.getFieldSpline(f), gen.V.grids[f]) for f in range(2)]
    return gen, kxs


@pytest.mark.parametrize("kind,degs,nels", [("RT", (1, 1), (5, 4)), ("RT", (2, 2), (4, 6)), ("N", (1, 1), (3, 5)), ("RT", (3, 3), (5, 3)),
                                            ("N", (2, 2), (6, 4)), ("RT", (1, 1), (1, 2))])
def test_2d_walks_with_different_bases_on_rows_and_columns(kind, degs, nels):
    """K_fg = M_f^T A_fg M_g for the components of a 2-D compatible B-spline (tIGAr/compatibleSplines.py:21-66): both fields
    extract to one Q_P grid (P = base degree + 1), the bases differ in degree per direction.  ``TensorPtAP2D.for_pair``: the
    two line walks with separate row- and column-side weights (padded to P + 1 functions per element), the last pass writing
    the true pattern -- against scipy's product of the Kronecker operators; pattern = the structural product."""
    import scipy.sparse as sps
    from tigar_amd import device as dev, forms as F
    from tigar_amd.tensorptap import TensorPtAP2D
    dev.device_info()
    gen, kxs = _compat_fields_2d(kind, degs, nels)
    g = gen.V.grids[0]
    assert g.degree == max(degs) + 1
    V1 = type(gen.V)([g], gen.V.element)
    A = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
    rng = np.random.default_rng(11)
    A.data = A.data + 0.3 * rng.standard_normal(A.nnz)
    Ad = dev.DeviceCSR.from_scipy(A)
    Ms = [sps.kron(kx.M1[1], kx.M1[0]).tocsr() for kx in kxs]
    for Mk in Ms:
        Mk.eliminate_zeros()
        Mk.sort_indices()
    dev.prof_reset()
    ones = lambda X: sps.csr_matrix((np.ones(X.nnz), X.indices, X.indptr), shape=X.shape)
    for f in range(2):
        for gg in range(2):
            plan = TensorPtAP2D.for_pair(kxs[f], kxs[gg])
            assert plan is not None
            K = plan.ptap(Ad)
            assert K is not None
            K = K.to_scipy()
            Kr = (Ms[f].T @ A @ Ms[gg]).tocsr()
            S = (ones(Ms[f]).T @ ones(A) @ ones(Ms[gg])).tocsr()
            S.sort_indices()
            assert K.shape == Kr.shape
            assert np.array_equal(K.indptr, S.indptr) and np.array_equal(K.indices, S.indices)
            assert abs(K - Kr).max() <= 1e-12 * abs(Kr).max()
    assert dev.prof_get(5)[1] >= 4
    # a block off the element-coupling pattern is declined (the caller takes the general kernels)
    Ab = A.tolil()
    Ab[0, A.shape[1] - 1] = 1.0
    assert TensorPtAP2D.for_pair(kxs[0], kxs[1]).ptap(dev.DeviceCSR.from_scipy(Ab.tocsr())) is None


@pytest.mark.parametrize("degs,nels", [((1, 1), (7, 5)), ((2, 2), (4, 5))])
def test_extract_matrix_on_a_2d_compatible_spline_takes_the_pair_walks(degs, nels, monkeypatch):
    """extractMatrix on a 2-D BSplineCompat("RT") space (demos/taylor-green/taylor-green-2d.py:68) with an assembled block matrix
    on the common Q_P grid: all four blocks through the 2-D walks with different row / column bases, K against the oracle
    (pattern and values, normal-direction boundary conditions), and the same with the walks switched off."""
    import scipy.sparse as sps
    import tigar_amd as t
    from tigar_amd import device as dev, forms as F, BSplines as B, common as tc
    from tigar_amd.compatibleSplines import BSplineCompat
    kv = [B.uniformKnots(degs[k], 0., 1. + 0.25 * k, nels[k]) for k in range(2)]
    gen = BSplineCompat(tc.selfcomm, B.ExplicitBSplineControlMesh(list(degs), kv), "RT", list(degs))
    for field in range(2):
        sp_f = gen.getFieldSpline(field)
        for side in (0, 1):
            gen.addZeroDofs(field, sp_f.getSideDofs(field, side))
    blocks = []
    for i in range(2):
        f = gen.getFieldSpline(i)
        blocks.append(O.generate_M_tensor(O.BSpline([s1.p for s1 in f.splines], [np.asarray(s1.knots) for s1 in f.splines])))
    Mo = sps.block_diag(blocks, format="csr")
    assert gen._kron_fields is not None and len(gen._kron_fields) == 2
    spline = t.ExtractedSpline(gen, 2 * (max(degs) + 1))
    g = spline.V.grids[0]
    V1 = type(spline.V)([g], spline.V.element)
    L = F.LaplaceForm().assemble_matrix(V1).to_scipy().tocsr()
    rng = np.random.default_rng(3)
    blk = [[L.copy() for _ in range(2)] for _ in range(2)]
    for r in blk:
        for Bm in r:
            Bm.data = Bm.data + 0.2 * rng.standard_normal(Bm.nnz)
    A = sps.bmat(blk, format="csr")
    zd = [int(i) for i in gen.zeroDofsArray()]
    dev.prof_reset()
    K = spline.extractMatrix(A, diag=2.5).to_scipy()
    assert dev.prof_get(5)[1] == 4                       # four blocks, four final passes of the walks
    Kr = O.extract_matrix(Mo, A, zd, diag=2.5)
    assert np.array_equal(K.indptr, Kr.indptr) and np.array_equal(K.indices, Kr.indices)
    assert abs(K - Kr).max() <= 1e-12 * abs(Kr).max()
    monkeypatch.setenv("TIGAR_PTAP_TENSOR", "0")
    K3 = spline.extractMatrix(A, diag=2.5).to_scipy()
    assert np.array_equal(K3.indices, Kr.indices) and abs(K3 - Kr).max() <= 1e-12 * abs(Kr).max()
