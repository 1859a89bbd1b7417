"""-m gpu: extractMatrix on cell-local FE spaces (csrc/tg_ptap_wave.hip, cell-block product; tigar_amd/cellptap.py): the
meshes of disconnected cells the reference builds for T-splines and multi-patch B-splines (tIGAr/RhinoTSplines.py:195-240)
make an assembled A block diagonal with dense blocks, and K = M^T A M (tIGAr/common.py:1194-1195) becomes a sum of small
dense triple products.  Against the oracle's / scipy's product: structural pattern identical, values to rounding,
MatZeroRowsColumns, bit-reproducibility, plan reuse, and the fall-back for matrices that are not of that kind."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tigar_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _cells(rng, ncell, b, ncp, nf_lo, nf_hi, zeros=0.15):
    """M of a cell-local space: cell c names nf_c random functions, its b rows hold values on (most of) them"""
    rows, cols, vals = [], [], []
    for c in range(ncell):
        nf = int(rng.integers(nf_lo, nf_hi + 1))
        funs = np.sort(rng.choice(ncp, size=nf, replace=False))
        for r in range(b):
            keep = rng.random(nf) > zeros
            keep[rng.integers(nf)] = True
            for f in funs[keep]:
                rows.append(c * b + r), cols.append(int(f)), vals.append(rng.standard_normal())
    M = sp.csr_matrix((vals, (rows, cols)), shape=(ncell * b, ncp))
    M.sort_indices()
    blocks = [rng.standard_normal((b, b)) for _ in range(ncell)]
    A = sp.block_diag(blocks, format="csr")
    A.sort_indices()
    return M, A


@pytest.mark.parametrize("ncell,b,ncp,nf_lo,nf_hi", [(200, 16, 300, 9, 25), (333, 9, 150, 4, 16), (50, 27, 400, 27, 64), (64, 4, 40, 1, 8)])
def test_cell_block_product_against_scipy(ncell, b, ncp, nf_lo, nf_hi):
    from tigar_amd.device import DeviceCSR
    from tigar_amd.cellptap import CellBlockPtAP, block_size_of
    rng = np.random.default_rng(ncell + b)
    M, A = _cells(rng, ncell, b, ncp, nf_lo, nf_hi)
    Md, Ad = DeviceCSR.from_scipy(M), DeviceCSR.from_scipy(A)
    assert block_size_of(Ad) == b
    plan = CellBlockPtAP(Md, b)
    zd = np.unique(rng.integers(0, ncp, size=7)).astype(np.int32)
    K = plan.ptap(Ad, zd, 2.5).to_scipy().tocsr()
    Ko = O.extract_matrix(M, A, list(zd), diag=2.5).tocsr()
    K.sort_indices(), Ko.sort_indices()
    # the structural pattern of the symbolic product (an entry wherever some M[r,i], A[r,s], M[s,j] are all stored), which
    # is what the oracle keeps
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # bit-reproducible, also when the rows are placed by the plan's row pointer (second call)
    K2 = plan.ptap(Ad, zd, 2.5).to_scipy().tocsr()
    K3 = CellBlockPtAP(Md, b).ptap(Ad, zd, 2.5).to_scipy().tocsr()
    for Kx in (K2, K3):
        assert np.array_equal(Kx.indices, K.indices) and np.array_equal(Kx.data.view(np.int64), K.data.view(np.int64))
    # new values on the same pattern through the same plan
    A2 = A.copy()
    A2.data = rng.standard_normal(A2.nnz)
    K4 = plan.ptap(DeviceCSR.from_scipy(A2), None).to_scipy()
    assert abs(K4 - (M.T @ A2 @ M)).max() <= 1e-12 * abs(M.T @ A2 @ M).max()
    # a matrix that is not block diagonal with dense blocks is declined (same entry count, one entry moved out of its block)
    A3 = A.tolil()
    A3[0, 0] = 0.0
    A3[0, b + 1] = 1.0
    A3 = A3.tocsr()
    A3.eliminate_zeros()
    assert A3.nnz == A.nnz and plan.ptap(DeviceCSR.from_scipy(A3), None) is None


def test_tspline_extract_matrix_takes_the_cell_block_product(monkeypatch):
    """through the API on the Rhino T-spline fixture: a block-diagonal A (what dolfin assembles on the mesh of disconnected
    cells) goes through the cell-block product, an arbitrary A through the general kernels -- the same K either way"""
    import tigar_amd as t
    from tigar_amd.RhinoTSplines import RhinoTSplineControlMesh
    gen = t.EqualOrderSpline(1, RhinoTSplineControlMesh(os.path.join(GOLDEN, "tspline_bicubic_patch.iga")))
    gen.addZeroDofs(0, [0, 1, 2, 29])
    spline = t.ExtractedSpline(gen, 6)
    rng = np.random.default_rng(8)
    A = sp.block_diag([rng.standard_normal((16, 16)) for _ in range(6)], format="csr")
    M = gen.M.to_scipy()
    K = spline.extractMatrix(A).to_scipy().tocsr()
    assert spline.__dict__.get("_cell_plans", {}).get(16) is not None            # the plan was built and used
    Ko = O.extract_matrix(M, A, [0, 1, 2, 29]).tocsr()
    K.sort_indices(), Ko.sort_indices()
    assert np.array_equal(K.indices, Ko.indices) and abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    monkeypatch.setenv("TIGAR_PTAP_CELLS", "0")
    Kg = spline.extractMatrix(A).to_scipy().tocsr()
    Kg.sort_indices()
    assert np.array_equal(Kg.indices, K.indices) and abs(Kg - K).max() <= 1e-12 * abs(Ko).max()


@pytest.mark.parametrize("ncell,b,ncp,nextra", [(300, 16, 260, 40), (500, 9, 200, 3), (120, 27, 350, 500)])
def test_cell_blocks_plus_couplings_outside_the_blocks(ncell, b, ncp, nextra, monkeypatch):
    """the reef-knot kind of matrix (demos/kl-shell-svk/reef-knot.py:455-467: a T-spline stiffness matrix plus
    contact terms added by hand -- the reason extractMatrix takes any A, tIGAr/common.py:1175).  The device splits A into its
    dense cell blocks and the remainder, the blocks go through the cell-block product, the remainder through the general
    kernels, the results are added on the union pattern: K against the oracle (structural pattern and values), a penalty
    of 1e9 among the extras, zero dofs; the same K as with the cell path switched off; plan reuse across calls."""
    import tigar_amd as t
    from tigar_amd import device as dev, common as tc
    from tigar_amd.device import DeviceCSR
    from tigar_amd.cellptap import split_cells, cell_size_with_extras
    rng = np.random.default_rng(ncell + nextra)
    M, A = _cells(rng, ncell, b, ncp, 4, min(40, ncp // 3))
    n = A.shape[0]
    r, c = rng.integers(0, n, nextra), rng.integers(0, n, nextra)
    keep = (r // b) != (c // b)
    v = rng.standard_normal(nextra)
    v[0] = 1e9                                                       # a penalty term
    E = sp.csr_matrix((v[keep], (r[keep], c[keep])), shape=(n, n))
    E.sum_duplicates()
    Ax = (A + E).tocsr()
    Ax.sort_indices()
    Ad = DeviceCSR.from_scipy(Ax)
    assert cell_size_with_extras(Ad) == b
    D, R = split_cells(Ad, b)
    assert abs(D.to_scipy() - A).max() == 0.0 and abs(R.to_scipy() - E).max() == 0.0
    # through the API: an ExtractedSpline whose M is this cell-local operator
    spline = t.ExtractedSpline.__new__(t.ExtractedSpline)
    spline.M, spline.MT = DeviceCSR.from_scipy(M), DeviceCSR.from_scipy(M.T.tocsr())
    spline.M._T = spline.MT
    spline.nFields, spline.comm, spline._kron, spline._kron_scalar, spline._kron_fields = 1, tc.selfcomm, None, None, None
    spline._ptap_plan = spline._ptap_plan_key = spline._slab = None
    zd = np.unique(rng.integers(0, ncp, size=5)).astype(np.int32)
    spline.zeroDofs = zd
    K = spline.extractMatrix(Ad, diag=3.0).to_scipy().tocsr()
    assert spline.__dict__.get("_cellR_key") is not None               # the split path ran
    Ko = O.extract_matrix(M, Ax, list(zd), diag=3.0).tocsr()
    K.sort_indices(), Ko.sort_indices()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    # entries away from the penalty keep their own accuracy (the two parts are summed separately)
    small = abs(Ko.data) < 1e3
    assert np.max(np.abs(K.data[small] - Ko.data[small])) <= 1e-10 * max(1.0, np.max(np.abs(Ko.data[small])))
    K2 = spline.extractMatrix(Ad, diag=3.0).to_scipy().tocsr()          # plans reused
    assert np.array_equal(K2.data.view(np.int64), K.data.view(np.int64))
    monkeypatch.setenv("TIGAR_PTAP_CELLS", "0")
    Kg = spline.extractMatrix(Ad, diag=3.0).to_scipy().tocsr()
    Kg.sort_indices()
    assert np.array_equal(Kg.indices, K.indices) and abs(Kg - K).max() <= 1e-12 * abs(Ko).max()
    # a matrix whose cell blocks are incomplete is declined by the split and still extracted
    monkeypatch.delenv("TIGAR_PTAP_CELLS")
    A3 = Ax.tolil()
    A3[1, 0] = 0.0
    A3 = A3.tocsr()
    A3.eliminate_zeros()
    assert split_cells(DeviceCSR.from_scipy(A3), b) is None
    K3 = spline.extractMatrix(DeviceCSR.from_scipy(A3), diag=3.0).to_scipy().tocsr()
    K3o = O.extract_matrix(M, A3, list(zd), diag=3.0).tocsr()
    K3.sort_indices(), K3o.sort_indices()
    assert np.array_equal(K3.indices, K3o.indices) and abs(K3 - K3o).max() <= 1e-12 * abs(K3o).max()
