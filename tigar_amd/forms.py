"""
FE-side INPUT generators for tensor-product patches with identity geometry (parametric ==
physical, as with ``ExplicitBSplineControlMesh``).  In the reference these matrices and
vectors come from ``dolfin.assemble(form)`` (tIGAr/common.py:1162-1220), which is outside
the extraction hot path; this module produces the same objects for the Q_p Lagrange space
on the knot mesh so that the path can be driven (and benchmarked) without FEniCS:

    LaplaceForm()            a(u,v) = (grad u, grad v)     -> K1xM1xM1 + M1xK1xM1 + M1xM1xK1
    MassForm()               a(u,v) = (u, v)
    SeparableLoadForm(f1d)   L(v)   = (f, v) with f(x) = scale * prod_k f1d[k](x_k)

Quadrature: Gauss-Legendre with p+1 points per direction (what ``quadrature_degree = 2p``
selects on quads/hexes [ext], demos/poisson/poisson.py:89).  The d-dimensional objects are
expanded on the GPU (``tg_kron_sum_csr`` / ``tg_vec_tensor3``); they are NOT timed as part
of the hot path.
"""
import numpy
import scipy.sparse as sp

from . import device as _dev


def _gauss01(n):
    x, w = numpy.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _lagrange01(p, t):
    nodes = numpy.arange(p + 1) / float(p)
    phi = numpy.ones((p + 1, len(t)))
    dphi = numpy.zeros((p + 1, len(t)))
    for a in range(p + 1):
        others = [m for m in range(p + 1) if m != a]
        for m in others:
            phi[a] *= (t - nodes[m]) / (nodes[a] - nodes[m])
        for m in others:
            term = numpy.full(len(t), 1.0 / (nodes[a] - nodes[m]))
            for q in others:
                if q != m:
                    term *= (t - nodes[q]) / (nodes[a] - nodes[q])
            dphi[a] += term
    return phi, dphi


def fe_matrices_1d(vertices, p, nq=None):
    """1-D CG Lagrange degree-p (mass, stiffness) on the mesh with the given vertices, as
    scipy CSR with identical (element-coupling) patterns."""
    nq = p + 1 if nq is None else nq
    t, w = _gauss01(nq)
    phi, dphi = _lagrange01(p, t)
    me = (phi * w) @ phi.T
    ke = (dphi * w) @ dphi.T
    v = numpy.asarray(vertices, dtype=numpy.float64)
    h = numpy.diff(v)
    nel = len(h)
    n = nel * p + 1
    loc = numpy.arange(p + 1)
    rows = (numpy.arange(nel)[:, None, None] * p + loc[None, :, None]) + 0 * loc[None, None, :]
    cols = (numpy.arange(nel)[:, None, None] * p + loc[None, None, :]) + 0 * loc[None, :, None]
    mvals = me[None, :, :] * h[:, None, None]
    kvals = ke[None, :, :] / h[:, None, None]
    Mm = sp.coo_matrix((mvals.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()
    Km = sp.coo_matrix((kvals.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()
    # force one shared structural pattern (coo->csr summation keeps explicit zeros)
    Mm.sort_indices()
    Km.sort_indices()
    return Mm, Km


def fe_load_1d(vertices, p, f, nq=None):
    nq = p + 1 if nq is None else nq
    t, w = _gauss01(nq)
    phi, _ = _lagrange01(p, t)
    v = numpy.asarray(vertices, dtype=numpy.float64)
    nel = len(v) - 1
    b = numpy.zeros(nel * p + 1)
    for e in range(nel):
        h = v[e + 1] - v[e]
        b[e * p:e * p + p + 1] += (phi * w) @ f(v[e] + h * t) * h
    return b


def _single_grid(V):
    if len(V.grids) != 1:
        raise NotImplementedError("synthetic forms are provided for single-field spaces")
    g = V.grids[0]
    if g.dg:
        raise NotImplementedError("synthetic forms are provided for CG spaces")
    return g


class LaplaceForm(object):
    """a(u,v) = int grad u . grad v on the parametric box."""

    def factors(self, V):
        g = _single_grid(V)
        mk = [fe_matrices_1d(g.vertices[k], g.degree) for k in range(g.dim())]
        d = g.dim()
        return [[mk[k][1] if k == dd else mk[k][0] for k in range(d)] for dd in range(d)]

    def assemble_matrix(self, V, row0=None, row1=None):
        return _dev.kron_sum_csr(self.factors(V), row0, row1)


class MassForm(object):
    """a(u,v) = int u v."""

    def factors(self, V):
        g = _single_grid(V)
        return [[fe_matrices_1d(g.vertices[k], g.degree)[0] for k in range(g.dim())]]

    def assemble_matrix(self, V, row0=None, row1=None):
        return _dev.kron_sum_csr(self.factors(V), row0, row1)


class SeparableLoadForm(object):
    """L(v) = int f v with f = scale * prod_k f1d[k](x_k)."""

    def __init__(self, f1d, scale=1.0):
        self.f1d, self.scale = list(f1d), float(scale)

    def vectors_1d(self, V):
        g = _single_grid(V)
        return [fe_load_1d(g.vertices[k], g.degree, self.f1d[k]) for k in range(g.dim())]

    def assemble_vector(self, V, row0=None, row1=None):
        return _dev.vec_tensor3(self.vectors_1d(V), self.scale, row0, row1)


class Equation(object):
    """``lhs == rhs`` stand-in for ufl.equation.Equation (tIGAr/common.py:1277-1283)."""

    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = lhs, rhs
