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
