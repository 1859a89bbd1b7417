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


def fe_gradient_1d(vertices, p, nq=None):
    """G[a,b] = int phi_a' phi_b of the 1-D CG Lagrange degree-p basis (scipy CSR on the element-coupling pattern of
    ``fe_matrices_1d``, explicit zeros kept): the mixed first-derivative factor of vector-valued forms."""
    nq = p + 1 if nq is None else nq
    t, w = _gauss01(nq)
    phi, dphi = _lagrange01(p, t)
    ge = (dphi * w) @ phi.T                          # (the 1/h of the derivative and the h of dx cancel)
    nel = len(vertices) - 1
    n = nel * p + 1
    loc = numpy.arange(p + 1)
    rows = (numpy.arange(nel)[:, None, None] * p + loc[None, :, None]) + 0 * loc[None, None, :]
    cols = (numpy.arange(nel)[:, None, None] * p + loc[None, None, :]) + 0 * loc[None, :, None]
    G = sp.coo_matrix((numpy.tile(ge, (nel, 1, 1)).ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()
    G.sort_indices()
    return G


def fe_matrices_1d_ext(vertices, p, nq=None):
    """(mass, stiffness, S2[a,b] = int phi_a'' phi_b'', C[a,b] = int phi_a'' phi_b) element by
    element (no inter-element terms, as dolfin assembles ``inner(lap(u),lap(v))*dx``), all on one
    shared pattern."""
    nq = p + 1 if nq is None else nq
    t, w = _gauss01(nq)
    phi, dphi = _lagrange01(p, t)
    # second derivatives of the equispaced Lagrange basis by differentiating the interpolant
    nodes = numpy.arange(p + 1) / float(p)
    V = numpy.vander(nodes, p + 1, increasing=True)          # monomial coefficients of each basis fn
    coef = numpy.linalg.solve(V, numpy.eye(p + 1))           # coef[:, a] = monomial coeffs of phi_a
    d2 = numpy.zeros((p + 1, len(t)))
    for a in range(p + 1):
        for m in range(2, p + 1):
            d2[a] += coef[m, a] * m * (m - 1) * t ** (m - 2)
    me = (phi * w) @ phi.T
    ke = (dphi * w) @ dphi.T
    s2e = (d2 * w) @ d2.T
    ce = (d2 * w) @ phi.T
    v = numpy.asarray(vertices, dtype=numpy.float64)
    h = numpy.diff(v)
    nel = len(h)
    n = nel * p + 1
    loc = numpy.arange(p + 1)
    rows = (numpy.arange(nel)[:, None, None] * p + loc[None, :, None]) + 0 * loc[None, None, :]
    cols = (numpy.arange(nel)[:, None, None] * p + loc[None, None, :]) + 0 * loc[None, :, None]

    def asm(e, scale):
        A = sp.coo_matrix(((e[None, :, :] * scale[:, None, None]).ravel(), (rows.ravel(), cols.ravel())),
                          shape=(n, n)).tocsr()
        A.sort_indices()
        return A
    return asm(me, h), asm(ke, 1.0 / h), asm(s2e, 1.0 / h ** 3), asm(ce, 1.0 / h)


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


def _memo(obj, V, build):
    """1-D factors depend only on the space: build once per (form, V)"""
    cache = obj.__dict__.setdefault("_cache", {})
    # keyed on what the factors are computed from (id(V) can be recycled after garbage collection)
    key = tuple((g.degree, bool(g.dg), tuple(numpy.asarray(v, dtype=numpy.float64).tobytes() for v in g.vertices))
                if hasattr(g, "vertices") else id(g) for g in V.grids)
    if key not in cache:
        cache[key] = build()
    return cache[key]


def _plane_window(g, row0, row1):
    """node planes [za, zb) of the last direction that hold the FE rows [row0, row1), and the planes [fa, fb) of the
    elements that touch them (whose control values the assembly reads)"""
    shape = g.shape()
    plane = int(numpy.prod(shape[:-1], dtype=numpy.int64)) if len(shape) > 1 else 1
    n_last, p = shape[-1], int(g.degree)
    nel = (n_last - 1) // p
    za, zb = int(row0) // plane, -(-int(row1) // plane)
    if zb <= za:
        return plane, za, zb, za, za
    e0 = za // p - 1 if (za > 0 and za % p == 0) else za // p
    e0 = min(e0, nel - 1)
    e1 = min(nel, (zb - 1) // p + 1)
    return plane, za, zb, e0 * p, e1 * p + 1


def _control_window(geometry, plane, fa, fb):
    """(the nsd+1 control functions as DeviceVectors, index of their first node): the functions themselves when they
    are held in full, else their values on the node planes [fa, fb) -- a rank of a distributed run holds only the rows
    it owns, and its forms read a window around them (``AbstractExtractionGenerator.controlFunctionWindow``)"""
    cpf = geometry.cpFuncs
    if all(getattr(fn, "local_range", None) is None for fn in cpf):
        return [fn.vector() for fn in cpf], 0
    gen = getattr(geometry, "_generator", geometry)
    return gen.controlFunctionWindow(fa, fb), fa * plane


def _mapped(geometry, V, form, row0=None, row1=None, grid=None, block=None):
    """dolfin.assemble stand-in on a mapped patch: ``geometry`` is a generator / ExtractedSpline
    whose ``cpFuncs`` (nsd+1 homogeneous control functions on the FE nodes of ``V_control``) define
    F = cpFuncs[i]/cpFuncs[nsd] (tIGAr/common.py:917-921); metric-based measure and gradient.  ``row0, row1``: the FE
    rows of a block only (global columns), as the z-slab pipeline asks for them.  ``form`` "elasticity": ``block`` =
    (i, j, lambda, mu), ``grid`` the node grid the fields share."""
    g = _single_grid(V) if grid is None else grid
    gc = _single_grid(geometry.V_control)
    if gc.shape() != g.shape():
        raise ValueError("the geometry lives on a different node grid than the space")
    verts = [g.vertices[k] for k in range(g.dim())]
    n = g.num_nodes()

    def assemble(cp, **rows):
        if form == "elasticity":
            if len(cp) - 1 != g.dim():
                raise ValueError("the elasticity form needs as many physical as parametric directions")
            return _dev.assemble_mapped_elasticity_block(verts, g.degree, cp, block[0], block[1], block[2], block[3], **rows)
        return _dev.assemble_mapped_matrix(verts, g.degree, cp, form, **rows)

    row0, row1 = (0 if row0 is None else int(row0)), (n if row1 is None else int(row1))       # (either may be left out)
    if (row0, row1) == (0, n):
        cp, node0 = _control_window(geometry, 1, 0, g.shape()[-1])
        if node0 == 0 and all(v.size() == n for v in cp):
            return assemble(cp)
        row0, row1 = 0, n
    plane, za, zb, fa, fb = _plane_window(g, row0, row1)
    cp, node0 = _control_window(geometry, plane, fa, fb)
    A = assemble(cp, row0=za * plane, row1=zb * plane, cp_node0=node0)
    if (za * plane, zb * plane) != (int(row0), int(row1)):           # (a range that cuts through node planes)
        A = A.block(int(row0) - za * plane, int(row1) - za * plane, 0, n)
    return A


class LaplaceForm(object):
    """a(u,v) = int grad u . grad v: on the parametric box (Kronecker sum of 1-D factors), or --
    with ``geometry`` (a generator or ExtractedSpline) -- in physical space on the mapped patch,
    grad and dx as ``spline.grad`` / ``spline.dx`` (tIGAr/common.py:917-945)."""

    def __init__(self, geometry=None):
        # a(u, v) = a(v, u) for THIS class; a subclass that adds terms says so itself (not inherited)
        self.symmetric = type(self) is LaplaceForm
        self.geometry = geometry

    def factors(self, V):
        def build():
            g = _single_grid(V)
            mk = [fe_matrices_1d(g.vertices[k], g.degree) for k in range(g.dim())]
            d = g.dim()
            return [[mk[k][1] if k == dd else mk[k][0] for k in range(d)] for dd in range(d)]
        return _memo(self, V, build)

    def assemble_matrix(self, V, row0=None, row1=None):
        if self.geometry is not None:
            return _mapped(self.geometry, V, "laplace", row0, row1)
        return _dev.kron_sum_csr(self.factors(V), row0, row1)


class ElasticityForm(object):
    """a(u,v) = int lambda div u div v + 2 mu eps(u):eps(v) on the parametric box of a d-field space (all fields on one
    CG node grid, dofs field after field -- the space of ``EqualOrderSpline(d, ...)``):
    block (i, j) = lambda int d_i phi_a d_j phi_b + mu int d_j phi_a d_i phi_b + delta_ij mu int grad phi_a . grad phi_b,
    each a Kronecker sum of 1-D factors (mass M, stiffness K, G[a,b] = int phi_a' phi_b) on the element-coupling pattern,
    written block by block by the Kronecker-sum kernel and put together on the device.  What a dolfin user writes as
    ``inner(sigma(u), eps(v))*dx`` for ``demos``-style linear elasticity on an identity-geometry patch.

    With ``geometry`` (a generator or ExtractedSpline with nsd == d): in physical space on the mapped patch --
    ``inner(sigma(u), spline.sym(spline.grad(v)))*spline.dx`` with the Cartesian derivatives of ``spline.grad`` /
    ``spline.div`` (tIGAr/common.py:1022-1040, calculusUtils.py:255-276); each block from the element kernels of
    csrc/tg_assemble.hip (no Kronecker factors: the PtAP takes the assembled blocks)."""

    def __init__(self, lmbda=1.0, mu=1.0, geometry=None):
        # a(u, v) = a(v, u) for THIS class; a subclass that adds terms says so itself (not inherited)
        self.symmetric = type(self) is ElasticityForm
        self.lmbda, self.mu = float(lmbda), float(mu)
        self.geometry = geometry

    def _grid(self, V):
        g = V.grids[0]
        if g.dg or len(V.grids) != g.dim() or any(
                gi.degree != g.degree or any(not numpy.array_equal(a, b) for a, b in zip(gi.axes, g.axes)) for gi in V.grids):
            raise NotImplementedError("ElasticityForm: as many fields as parametric directions, all on one CG node grid")
        return g

    def block_factors(self, V):
        """factors[i][j] = list of terms, each a list of d 1-D matrices (direction 0 first); None on a mapped patch"""
        if self.geometry is not None:
            return None

        def build():
            g = self._grid(V)
            d = g.dim()
            one = [fe_matrices_1d(g.vertices[k], g.degree) + (fe_gradient_1d(g.vertices[k], g.degree),) for k in range(d)]
            Mk, Kk, Gk = [o[0] for o in one], [o[1] for o in one], [o[2] for o in one]
            lam, mu = self.lmbda, self.mu
            out = [[None] * d for _ in range(d)]
            for i in range(d):
                for j in range(d):
                    if i == j:
                        terms = []
                        for k in range(d):
                            c = (lam + 2.0 * mu) if k == i else mu
                            terms.append([(c * Kk[q]) if q == k else Mk[q] for q in range(d)])
                    else:
                        # int d_i phi_a d_j phi_b: G in direction i, G^T in direction j; and the transposed pairing
                        t1 = [(lam * Gk[q]) if q == i else (Gk[q].T.tocsr() if q == j else Mk[q]) for q in range(d)]
                        t2 = [(mu * Gk[q].T.tocsr()) if q == i else (Gk[q] if q == j else Mk[q]) for q in range(d)]
                        terms = [t1, t2]
                    out[i][j] = terms
            return out
        return _memo(self, V, build)

    def assemble_block(self, V, i, j, row0=None, row1=None):
        """rows [row0, row1) of block (i, j) (fields i, j; columns of one field)"""
        if self.geometry is not None:
            return _mapped(self.geometry, V, "elasticity", row0, row1, grid=self._grid(V), block=(i, j, self.lmbda, self.mu))
        return _dev.kron_sum_csr(self.block_factors(V)[i][j], row0, row1)

    def assemble_matrix(self, V, row0=None, row1=None):
        if row0 is not None or row1 is not None:
            raise NotImplementedError("row blocks of the elasticity form: use assemble_block")
        if self.geometry is not None:
            d = self._grid(V).dim()
            return _dev.csr_from_blocks([[self.assemble_block(V, i, j) for j in range(d)] for i in range(d)])
        fac = self.block_factors(V)
        d = len(fac)
        return _dev.csr_from_blocks([[_dev.kron_sum_csr(fac[i][j]) for j in range(d)] for i in range(d)])


class MassForm(object):
    """a(u,v) = int u v (``geometry``: see LaplaceForm)."""

    def __init__(self, geometry=None):
        # a(u, v) = a(v, u) for THIS class; a subclass that adds terms says so itself (not inherited)
        self.symmetric = type(self) is MassForm
        self.geometry = geometry

    def factors(self, V):
        g = _single_grid(V)
        return [[fe_matrices_1d(g.vertices[k], g.degree)[0] for k in range(g.dim())]]

    def assemble_matrix(self, V, row0=None, row1=None):
        if self.geometry is not None:
            return _mapped(self.geometry, V, "mass", row0, row1)
        return _dev.kron_sum_csr(self.factors(V), row0, row1)


class NodalLoadForm(object):
    """L(v) = int f_h v dx on the mapped patch, f_h = nodal interpolant of ``f``: a callable evaluated at the
    physical node positions x = F(node) (array of shape [nnodes, nsd] -> values), a number (constant load), or the node
    values themselves (array / DeviceVector on all FE nodes)."""

    def __init__(self, f, geometry):
        self.f, self.geometry = f, geometry

    def _nodal_values(self, cp, node0, n_nodes):
        """f on the nodes [node0, node0 + n_nodes) that the control functions ``cp`` are given on"""
        if callable(self.f):
            # f at the physical node positions x = F(node): evaluated on the host (f is the user's Python function), once per
            # window of control functions -- the sub-slabs of one assembly and the assemblies of a Newton iteration ask for the
            # same windows again (keyed on the vectors' identity: a new geometry means new control functions)
            key = (tuple(id(v) for v in cp), int(node0), int(n_nodes))
            cache = self.__dict__.setdefault("_fnodal", {})
            if key not in cache:
                if len(cache) > 64:
                    cache.clear()
                c = [v.get_local() for v in cp]
                x = numpy.stack([c[i] / c[-1] for i in range(len(c) - 1)], axis=1)
                cache[key] = (_dev.DeviceVector(data=numpy.asarray(self.f(x), dtype=numpy.float64)), cp)   # (cp kept alive: ids)
            return cache[key][0]
        if numpy.isscalar(self.f):
            fn = _dev.DeviceVector(n_nodes, zero=False)
            fn.fill(float(self.f))
            return fn
        full = self.f if isinstance(self.f, _dev.DeviceVector) else _dev.DeviceVector(data=self.f)
        if node0 == 0 and full.size() == n_nodes:
            return full
        piece = _dev.DeviceVector(n_nodes, zero=False)
        _dev.vec_copy_range(piece, 0, full, node0, n_nodes)
        return piece

    def assemble_vector(self, V, row0=None, row1=None):
        g = _single_grid(V)
        verts = [g.vertices[k] for k in range(g.dim())]
        n = g.num_nodes()
        row0, row1 = (0 if row0 is None else int(row0)), (n if row1 is None else int(row1))   # (either may be left out)
        if (row0, row1) == (0, n):
            cp, node0 = _control_window(self.geometry, 1, 0, g.shape()[-1])
            if node0 == 0 and all(v.size() == n for v in cp):
                return _dev.assemble_mapped_load(verts, g.degree, cp, self._nodal_values(cp, 0, n))
            row0, row1 = 0, n
        plane, za, zb, fa, fb = _plane_window(g, row0, row1)
        cp, node0 = _control_window(self.geometry, plane, fa, fb)
        if node0 == 0 and cp[0].size() == n:
            fn = self._nodal_values(cp, 0, n)
        else:
            fn = self._nodal_values(cp, node0, cp[0].size())
        b = _dev.assemble_mapped_load(verts, g.degree, cp, fn, row0=za * plane, row1=zb * plane, cp_node0=node0)
        if (za * plane, zb * plane) != (int(row0), int(row1)):
            piece = _dev.DeviceVector(int(row1) - int(row0), zero=False)
            _dev.vec_copy_range(piece, 0, b, int(row0) - za * plane, int(row1) - int(row0))
            b = piece
        return b


class SeparableLoadForm(object):
    """L(v) = int f v with f = scale * prod_k f1d[k](x_k)."""

    def __init__(self, f1d, scale=1.0):
        self.f1d, self.scale = list(f1d), float(scale)

    def vectors_1d(self, V):
        def build():
            g = _single_grid(V)
            return [fe_load_1d(g.vertices[k], g.degree, self.f1d[k]) for k in range(g.dim())]
        return _memo(self, V, build)

    def assemble_vector(self, V, row0=None, row1=None):
        return _dev.vec_tensor3(self.vectors_1d(V), self.scale, row0, row1)


class BiharmonicForm(object):
    """a(u,v) = int (lap u)(lap v), element-wise (demos/biharmonic/biharmonic.py:100-103), 2-D:
    S2xM + MxS2 + C^T x C + C x C^T with C[a,b] = int phi_a'' phi_b.  With ``geometry`` (nsd == d, 2-D or 3-D): on the
    mapped patch, lap = spline.div(spline.grad(.)) with the second derivatives of the (rational) map
    (tIGAr/common.py:1022-1040; csrc/tg_assemble.hip, plain element kernel)."""

    def __init__(self, geometry=None):
        # a(u, v) = a(v, u) for THIS class; a subclass that adds terms says so itself (not inherited)
        self.symmetric = type(self) is BiharmonicForm
        self.geometry = geometry

    def factors(self, V):
        g = _single_grid(V)
        if g.dim() != 2:
            raise NotImplementedError("BiharmonicForm is provided for 2-D patches")
        (Mx, _, S2x, Cx), (My, _, S2y, Cy) = [fe_matrices_1d_ext(g.vertices[k], g.degree) for k in range(2)]
        return [[S2x, My], [Mx, S2y], [Cx.T.tocsr(), Cy], [Cx, Cy.T.tocsr()]]

    def assemble_matrix(self, V, row0=None, row1=None):
        if self.geometry is not None:
            return _mapped(self.geometry, V, "biharmonic", row0, row1)
        return _dev.kron_sum_csr(self.factors(V), row0, row1)


class SumOfSeparableLoads(object):
    """L(v) = int f v with f = sum_t scale_t * prod_k f1d_t[k](x_k)."""

    def __init__(self, terms):
        self.terms = [SeparableLoadForm(f1d, scale) for f1d, scale in terms]

    def assemble_vector(self, V, row0=None, row1=None):
        out = None
        for t in self.terms:
            v = t.assemble_vector(V, row0, row1)
            if out is None:
                out = v
            else:
                out.axpy(1.0, v)
        return out


class Equation(object):
    """``lhs == rhs`` stand-in for ufl.equation.Equation (tIGAr/common.py:1277-1283)."""

    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = lhs, rhs


class SemilinearResidual(object):
    """Residual of  -lap u + g(u) = f  in group-FE (product approximation) form on the scalar Q_p
    space:  R_a = sum_b K_ab u_b + sum_b M_ab (g(u_b) - f_b), with K, M the stiffness and mass
    matrices (identity geometry, or mapped with ``geometry``), ``u`` the current FE ``Function`` (read
    at assembly time, as a UFL form reads its coefficient), ``g`` / ``dg`` callables mapping a
    ``DeviceVector`` of nodal values to one (``lambda v: v.pointwise_mult(v).pointwise_mult(v)``).
    ``tangent()`` gives the matching Jacobian form  J = K + M diag(g'(u)).  K and M are assembled
    once and reused by every Newton step (fixed sparsity)."""

    def __init__(self, u, f_nodal, g, dg, geometry=None):
        self.u, self.g, self.dg, self.geometry = u, g, dg, geometry
        self.f = f_nodal if isinstance(f_nodal, _dev.DeviceVector) else _dev.DeviceVector(data=f_nodal)
        self._KM = None

    def _matrices(self, V):
        if self._KM is None:
            self._KM = (LaplaceForm(self.geometry).assemble_matrix(V), MassForm(self.geometry).assemble_matrix(V))
        return self._KM

    def _row_blocks(self, V, row0, row1):
        """rows [row0, row1) of K and M (cut once per row range: a Newton loop asks for the same blocks every step)"""
        key = (int(row0), int(row1))
        if getattr(self, "_blocks", None) is None or self._blocks[0] != key:
            K, Mm = self._matrices(V)
            n = K.shape[1]
            self._blocks = (key, K.block(key[0], key[1], 0, n), Mm.block(key[0], key[1], 0, n))
        return self._blocks[1], self._blocks[2]

    def assemble_vector(self, V, row0=None, row1=None):
        K, Mm = self._matrices(V)
        # (a rank-local u: its ghosted form -- own rows + the rows of the z-neighbours this rank's rows couple to)
        uv = self.u.ghosted() if hasattr(self.u, "ghosted") else self.u.vector()
        if row0 is not None and (row0, row1) != (0, K.shape[0]):
            K, Mm = self._row_blocks(V, row0, row1)
        r = K.mult(uv)
        t = self.g(uv)
        t.axpy(-1.0, self.f)
        r.axpy(1.0, Mm.mult(t))
        return r

    def tangent(self):
        return _SemilinearTangent(self)


class _SemilinearTangent(object):
    def __init__(self, res):
        self.res = res

    def assemble_matrix(self, V, row0=None, row1=None):
        K, Mm = self.res._matrices(V)
        uv = self.res.u.ghosted() if hasattr(self.res.u, "ghosted") else self.res.u.vector()
        if row0 is not None and (row0, row1) != (0, K.shape[0]):
            K, Mm = self.res._row_blocks(V, row0, row1)
        return K.combine(1.0, Mm, 1.0, self.res.dg(uv))
