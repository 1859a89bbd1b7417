"""
Extraction operators that are never materialised.

``generateM`` (tIGAr/common.py:1516-1578) stores M as a PETSc AIJ matrix.  For a tensor-product
B-spline on its Q_p node grid M is the Kronecker product of the 1-D extraction matrices whenever the
``abs(v) > eps`` filter (tIGAr/common.py:1569) dropped nothing but exact zeros -- checked, not assumed --
and at BASELINE cfg3 (256^3, p=3) its CSR form would need 271 GB (M) + 271 GB (M^T).  ``ImplicitExtraction``
stands in for the matrix object behind the unchanged generator / ``ExtractedSpline`` API: it knows its
shape and nnz, applies M and M^T by three 1-D passes (``tg_tensor_apply_1d``), materialises row ranges on
request (``rows``; the same ``k_extract_fill`` kernel as the resident path, so the entries are the
reference's bit for bit) and hands ``extractMatrix`` the 1-D factors for the sum-factorised PtAP.
"""
import numpy as np

from . import device as _dev


class LazyFEMatrix(object):
    """FE matrix given by a producer of row blocks instead of a resident CSR matrix: ``rows(r0, r1)``
    returns the global FE rows [r0, r1) as a ``DeviceCSR`` with global column indices.  What
    ``dolfin.assemble`` hands to ``extractMatrix`` in the reference (tIGAr/common.py:1206-1220) when the
    assembled matrix does not fit in HBM at once (cfg3: 684 GB)."""

    def __init__(self, producer, shape, kron_factors=None):
        self._producer = producer
        self.shape = (int(shape[0]), int(shape[1]))
        # the matrix as a Kronecker sum of 1-D matrices (factors[t][k], scipy CSR) when its producer is one -- the PtAP
        # can then form the entries inside its first pass instead of reading row blocks (never materialised)
        self.kron_factors = kron_factors
        # several fields: block_rows(f, g, r0, r1) -> rows of ONE field block with the columns of one field (or None when
        # the fields are not coupled), block_factors[f][g] -> that block as a Kronecker sum
        self.block_rows = None
        self.block_factors = None

    def rows(self, r0, r1):
        return self._producer(int(r0), int(r1))


class LazyFEVector(object):
    """FE vector given by a producer of entry ranges: ``rows(r0, r1)`` -> ``DeviceVector``."""

    def __init__(self, producer, n):
        self._producer = producer
        self.n = int(n)

    def size(self):
        return self.n

    def rows(self, r0, r1):
        return self._producer(int(r0), int(r1))


class ImplicitExtraction(object):
    """M = M_(d-1) (x) ... (x) M_0 of a tensor ``BSpline`` on its node grid, not stored.

    ``kx``: the ``KronExtraction`` (1-D factors); ``transposed``: this object stands for M^T."""

    is_implicit = True

    def __init__(self, kx, eps, transposed=False, _pair=None):
        self.kx, self.eps, self.transposed = kx, float(eps), bool(transposed)
        self._T = _pair
        nfe, ncp = int(np.prod(kx.nfe, dtype=np.int64)), int(np.prod(kx.ncp, dtype=np.int64))
        self._shape = (ncp, nfe) if transposed else (nfe, ncp)

    # ---- what the API reads from a matrix object ------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def nnz(self):
        return self.kx.nnz_product

    def transpose(self):
        if self._T is None:
            self._T = ImplicitExtraction(self.kx, self.eps, not self.transposed, _pair=self)
        return self._T

    def _apply(self, x, transposed, y=None):
        kx, d = self.kx, self.kx.d
        dims = list(kx.ncp if not transposed else kx.nfe)
        t = x
        for k in range(d):
            F = kx.M1T[k] if transposed else kx.M1[k]
            t = _dev.tensor_apply_1d(t, dims, k, F, out=y if k == d - 1 else None)
            dims[k] = F.shape[0]
        return t

    def mult(self, x, y=None):
        """y = M x (or M^T x for the transposed object): one 1-D pass per direction."""
        return self._apply(x, self.transposed, y)

    def mult_transpose(self, b, y=None):
        return self._apply(b, not self.transposed, y)

    def __mul__(self, x):
        if isinstance(x, _dev.DeviceVector):
            return self.mult(x)
        return NotImplemented

    # ---- materialisation ---------------------------------------------------------------------------
    def rows(self, r0, r1):
        """Rows [r0, r1) as a ``DeviceCSR`` (global columns): the extraction kernels on a row range."""
        if self.kx.columns_ascending():
            # (an implicit operator is an exact Kronecker product by construction: pencil walk, closed-form row starts)
            return _dev.kron3_csr(self.kx.M1T if self.transposed else self.kx.M1, r0, r1, 0, self._shape[1])
        sp1, axes = self.kx.basis.splines, self.kx.grid.axes
        if self.transposed:
            return _dev.extract_csr_tensor_t(sp1, axes, 0, self._shape[1], self.eps, r0, r1)
        return _dev.extract_csr_tensor(sp1, axes, 0, self._shape[1], self.eps, r0, r1)

    def materialise(self):
        need = 12.0 * self.nnz + 8.0 * self._shape[0]
        free_b = _dev.mem_info()[0] + _dev.pool_stats()[0]
        if need > 0.8 * free_b:
            raise MemoryError("the extraction operator needs %.0f GB as a CSR matrix (%.0f GB of HBM free); "
                              "use rows(r0, r1) for row ranges" % (need / 1e9, free_b / 1e9))
        return self.rows(0, self._shape[0])

    def to_scipy(self):
        return self.materialise().to_scipy()

    def rows_to_scipy(self, r0, r1):
        return self.rows(r0, r1).to_scipy()


class BlockImplicitExtraction(object):
    """M = diag(M_s, ..., M_s) of ``nfields`` fields on ONE tensor basis (``EqualOrderSpline(nFields > 1)``), dofs and FE
    rows field after field, M_s = the implicit scalar operator -- not stored.  Stands in for the matrix object when the
    patch is spread over several ranks (no rank holds all rows) or M would not fit."""

    is_implicit = True

    def __init__(self, kx, nfields, eps, transposed=False, _pair=None):
        self.kx, self.nfields, self.eps, self.transposed = kx, int(nfields), float(eps), bool(transposed)
        self.scalar = ImplicitExtraction(kx, eps, transposed)
        self._T = _pair
        r, c = self.scalar.shape
        self._shape = (self.nfields * r, self.nfields * c)

    @property
    def shape(self):
        return self._shape

    @property
    def nnz(self):
        return self.nfields * self.kx.nnz_product

    def transpose(self):
        if self._T is None:
            self._T = BlockImplicitExtraction(self.kx, self.nfields, self.eps, not self.transposed, _pair=self)
        return self._T

    def _apply(self, x, op, y=None):
        r, c = op.shape
        out = y if y is not None else _dev.DeviceVector(self.nfields * r)
        for f in range(self.nfields):
            xf = _dev.DeviceVector(c)
            _dev.vec_copy_range(xf, 0, x, f * c, c)
            yf = op.mult(xf)
            _dev.vec_copy_range(out, f * r, yf, 0, r)
        return out

    def mult(self, x, y=None):
        return self._apply(x, self.scalar, y)

    def mult_transpose(self, b, y=None):
        return self._apply(b, self.scalar.transpose(), y)

    def __mul__(self, x):
        if isinstance(x, _dev.DeviceVector):
            return self.mult(x)
        return NotImplemented

    def materialise(self):
        """the block-diagonal matrix as a DeviceCSR (tests, small patches)"""
        import scipy.sparse as sp
        blk = self.scalar.to_scipy()
        return _dev.DeviceCSR.from_scipy(sp.block_diag([blk] * self.nfields, format="csr"))

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.block_diag([self.scalar.to_scipy()] * self.nfields, format="csr")


class FieldListImplicitExtraction(object):
    """M = diag(M_0, ..., M_{nF-1}) of fields on DIFFERENT tensor bases (``FieldListSpline``, the components of a compatible
    B-spline: tIGAr/common.py:1949-1970, tIGAr/compatibleSplines.py:21-101), FE rows and dofs field after field, every M_f
    the implicit Kronecker operator of its own basis -- not stored.  Stands in for the matrix object when the patch is
    spread over several ranks (no rank holds all rows) or TIGAR_IMPLICIT_M=1 asks for it."""

    is_implicit = True

    def __init__(self, kxs, eps, transposed=False, _pair=None):
        self.kxs, self.eps, self.transposed = list(kxs), float(eps), bool(transposed)
        self.nfields = len(self.kxs)
        self.scalars = [ImplicitExtraction(kx, eps, transposed) for kx in self.kxs]
        self._T = _pair
        self._shape = (sum(s.shape[0] for s in self.scalars), sum(s.shape[1] for s in self.scalars))

    @property
    def shape(self):
        return self._shape

    @property
    def nnz(self):
        return sum(kx.nnz_product for kx in self.kxs)

    def transpose(self):
        if self._T is None:
            self._T = FieldListImplicitExtraction(self.kxs, self.eps, not self.transposed, _pair=self)
        return self._T

    def _apply(self, x, ops, y=None):
        out = y if y is not None else _dev.DeviceVector(sum(op.shape[0] for op in ops))
        r0 = c0 = 0
        for op in ops:
            r, c = op.shape
            xf = _dev.DeviceVector(c)
            _dev.vec_copy_range(xf, 0, x, c0, c)
            _dev.vec_copy_range(out, r0, op.mult(xf), 0, r)
            r0 += r
            c0 += c
        return out

    def mult(self, x, y=None):
        return self._apply(x, self.scalars, y)

    def mult_transpose(self, b, y=None):
        return self._apply(b, [s.transpose() for s in self.scalars], y)

    def __mul__(self, x):
        if isinstance(x, _dev.DeviceVector):
            return self.mult(x)
        return NotImplemented

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.block_diag([s.to_scipy() for s in self.scalars], format="csr")

    def materialise(self):
        return _dev.DeviceCSR.from_scipy(self.to_scipy())
