"""
The ``RhinoTSplines`` module of tigar_amd: T-splines given by element-wise Bezier extraction operators in the
text format of the Rhino T-splines plug-in, as read by ``tIGAr/RhinoTSplines.py`` (scalar basis :67-137,
control mesh :243-286).  Same classes and semantics; the extraction rows -- per Bezier element e and FE node,
N_a = sum_b C_e[a][b] B_b -- are evaluated by the HIP kernel ``tg_extract_csr_bezier`` (csrc/tg_bezier.hip) in
one batch instead of a Python call per FE node.

File layout (as the reference parses it): line 1 ``<kind> ncp``, line 2 ``<kind> nel``, ``ncp`` control-point
lines ``<tag> x y z w`` from line 3 on, then per element: a header whose second token is the number of
functions ``nshl``, a line with their global indices, and ``nshl`` lines of 16 Bernstein coefficients
(bicubic elements; Bernstein index i + 4 j).

FE side: the reference meshes every Bezier element as its own disconnected cell [3e, 3e+2] x [-1, 1]
(:139-228) with degree-3 Lagrange nodes (``getDegree`` :233-237); ``BezierElementNodeGrid`` is that node set.
"""
import math

import numpy

from .common import AbstractScalarBasis, AbstractControlMesh, USE_RECT_ELEM_DEFAULT, worldcomm
from . import device as _dev

ELEMENT_PITCH = 3.0          # x-distance between the origins of consecutive Bezier elements (:62-66)
ELEMENT_WIDTH = 2.0


def Bernstein_p3(u):
    """The four cubic Bernstein polynomials on (-1, 1) at ``u`` (tIGAr/RhinoTSplines.py:16-35; same expressions,
    so the values are the reference's bit for bit)."""
    x = 0.5 * (1.0 + u)
    omx = 1.0 - x
    return [omx ** 3, 3.0 * x * (omx ** 2), 3.0 * (x ** 2) * omx, x ** 3]


def RhinoTSplineScalarBasisFuncs(xi, C):
    """T-spline functions of one Bezier element at ``xi`` in (-1,1)^2 from its extraction rows ``C``
    (tIGAr/RhinoTSplines.py:37-60): host-side single-point form (the batch form runs on the GPU)."""
    Bu, Bv = Bernstein_p3(xi[0]), Bernstein_p3(xi[1])
    bern = [Bu[i] * Bv[j] for j in range(4) for i in range(4)]
    out = []
    for row in C:
        acc = 0.0
        for b in range(16):
            acc += row[b] * bern[b]
        out.append(acc)
    return out


class BezierElementNodeGrid(object):
    """Lagrange nodes of the reference's mesh of disconnected Bezier elements: element e occupies
    [3e, 3e+2] x [-1, 1]; (degree+1)^2 equispaced nodes each, x fastest; rows = element-major."""

    def __init__(self, nel, degree):
        self.nel, self.degree, self.dg = int(nel), int(degree), True
        t = numpy.arange(self.degree + 1, dtype=numpy.float64) / float(self.degree)
        self.t = t

    def dim(self):
        return 2

    def num_nodes(self):
        return self.nel * (self.degree + 1) ** 2

    def element_axes(self, e):
        """(x nodes, y nodes) of element e: x = x0 (1-t) + x1 t with the end points exact (as BSpline1.feNodes)"""
        x0, x1 = ELEMENT_PITCH * e, ELEMENT_PITCH * e + ELEMENT_WIDTH
        xs = x0 * (1.0 - self.t) + x1 * self.t
        xs[0], xs[-1] = x0, x1
        ys = -1.0 * (1.0 - self.t) + 1.0 * self.t
        ys[0], ys[-1] = -1.0, 1.0
        return xs, ys

    def coordinates(self):
        q = self.degree + 1
        out = numpy.empty((self.num_nodes(), 2))
        for e in range(self.nel):
            xs, ys = self.element_axes(e)
            out[e * q * q:(e + 1) * q * q, 0] = numpy.tile(xs, q)
            out[e * q * q:(e + 1) * q * q, 1] = numpy.repeat(ys, q)
        return out


def _tokens(line):
    return line.split()


class RhinoTSplineScalarBasis(AbstractScalarBasis):
    """Scalar T-spline basis from a Rhino element-extraction file (tIGAr/RhinoTSplines.py:67-237)."""

    def __init__(self, fname, useRect=USE_RECT_ELEM_DEFAULT):
        self.nvar = 2
        self.useRect = useRect
        with open(fname, "r") as f:
            lines = f.read().split("\n")
        self.ncp = int(_tokens(lines[1])[1])
        self.nelBez = int(_tokens(lines[2])[1])
        cursor = 3 + self.ncp                       # control points occupy lines 3 .. 3+ncp-1
        self.extractionNodes, self.extractionOperators = [], []
        self.maxNshl = 0
        for _ in range(self.nelBez):
            nshl = int(_tokens(lines[cursor])[1])
            self.maxNshl = max(self.maxNshl, nshl)
            self.extractionNodes.append([int(tok) for tok in _tokens(lines[cursor + 1])])
            rows = [[float(tok) for tok in _tokens(lines[cursor + 2 + a])] for a in range(nshl)]
            self.extractionOperators.append(rows)
            cursor += nshl + 2

    # ---- AbstractScalarBasis --------------------------------------------------------------------
    def getPrealloc(self):
        return self.maxNshl

    def useRectangularElements(self):
        return self.useRect

    def needsDG(self):
        return False

    def getNcp(self):
        return self.ncp

    def getDegree(self):
        return 3 if self.useRect else 6

    def elementFromCoordinates(self, xi):
        return int(xi[0] / ELEMENT_PITCH + 0.1)

    def getNodesAndEvals(self, xi):
        e = self.elementFromCoordinates(xi)
        u = xi[0] - ELEMENT_PITCH * e - 1.0
        vals = RhinoTSplineScalarBasisFuncs([u, xi[1]], self.extractionOperators[e])
        return [[node, val] for node, val in zip(self.extractionNodes[e], vals)]

    def generateMesh(self, comm=worldcomm, degree=None, dg=False):
        if not self.useRect:
            raise NotImplementedError("triangular extraction elements (useRect=False) are not provided")
        return BezierElementNodeGrid(self.nelBez, self.getDegree() if degree is None else degree)

    # ---- batch extraction on the device -----------------------------------------------------------
    def extractBlockOnDevice(self, grid, col_offset, ncols, eps):
        """All rows of this basis on ``grid`` as a DeviceCSR: per element the Bernstein values at its FE nodes
        (host, the reference's expressions and local coordinates u = x - 3e - 1) and the extraction rows sorted by
        function index; the contraction, the abs(v) > eps filter and the CSR build run in ``tg_extract_csr_bezier``."""
        q = grid.degree + 1
        nloc = q * q
        bern = numpy.empty((self.nelBez, nloc, 16))
        for e in range(self.nelBez):
            xs, ys = grid.element_axes(e)
            Bu = [Bernstein_p3(float(x) - ELEMENT_PITCH * self.elementFromCoordinates([float(x)]) - 1.0) for x in xs]
            Bv = [Bernstein_p3(float(y)) for y in ys]
            for jn in range(q):
                for i_n in range(q):
                    bern[e, jn * q + i_n, :] = [Bu[i_n][i] * Bv[jn][j] for j in range(4) for i in range(4)]
        eoff = numpy.zeros(self.nelBez + 1, dtype=numpy.int64)
        nodes, coef = [], []
        for e in range(self.nelBez):
            order = numpy.argsort(numpy.asarray(self.extractionNodes[e]), kind="stable")
            nd = numpy.asarray(self.extractionNodes[e], dtype=numpy.int64)[order]
            if numpy.any(numpy.diff(nd) == 0):
                raise ValueError("element %d lists a basis function twice" % e)
            nodes.append(nd)
            coef.append(numpy.asarray(self.extractionOperators[e], dtype=numpy.float64).reshape(len(nd), 16)[order])
            eoff[e + 1] = eoff[e] + len(nd)
        return _dev.extract_csr_bezier(bern, eoff, numpy.concatenate(nodes) if nodes else numpy.zeros(0),
                                       numpy.concatenate(coef) if coef else numpy.zeros((0, 16)), col_offset, ncols, eps)


class RhinoTSplineControlMesh(AbstractControlMesh):
    """Control mesh from the same file: homogeneous control points (w x, w y, w z, w)
    (tIGAr/RhinoTSplines.py:243-286)."""

    def __init__(self, fname, useRect=USE_RECT_ELEM_DEFAULT):
        self.scalarSpline = RhinoTSplineScalarBasis(fname, useRect)
        self.nsd = 3
        with open(fname, "r") as f:
            lines = f.read().split("\n")
        n = self.scalarSpline.getNcp()
        pts = numpy.array([[float(tok) for tok in _tokens(lines[3 + i])[1:self.nsd + 2]] for i in range(n)])
        self.bnet = pts.copy()
        self.bnet[:, :self.nsd] = pts[:, :self.nsd] * pts[:, self.nsd:self.nsd + 1]      # homogenise

    def getHomogeneousCoordinate(self, node, direction):
        return self.bnet[node, direction]

    def getHomogeneousCoordinates(self):
        return self.bnet

    def getScalarSpline(self):
        return self.scalarSpline

    def getNsd(self):
        return self.nsd
