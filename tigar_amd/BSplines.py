"""
The ``BSplines`` module of tigar_amd -- same public surface as the reference's
``tIGAr/BSplines.py`` (uniformKnots, BSpline1, BSpline, ExplicitBSplineControlMesh), with
the FE mesh replaced by an implicit tensor-product Q_p node grid and every basis-function
evaluation executed by the HIP kernels (device twin of ``basisFuncsInner``,
tIGAr/BSplines.py:73-120).  Host code here is bookkeeping only (knot vectors, index maps).
"""
import numpy

from .common import (AbstractScalarBasis, AbstractControlMesh, INDEX_TYPE,
                     USE_RECT_ELEM_DEFAULT, worldcomm, DOLFIN_EPS, near, TensorNodeGrid)
from . import device as _dev

# custom eps for checking knots (tIGAr/BSplines.py:42)
KNOT_NEAR_EPS = 10.0 * DOLFIN_EPS


def uniformKnots(p, start, end, N, periodic=False, continuityDrop=0):
    """
    Uniform open (or periodic) knot vector of degree ``p`` with ``N`` elements; interior
    knots have multiplicity ``continuityDrop+1``.  Same values as the reference
    (tIGAr/BSplines.py:14-38): interior knots are ``start + float(i)*h``.
    Raises ``ValueError`` where the reference prints an error and exits.
    """
    if continuityDrop >= p:
        raise ValueError("Continuity drop too high for spline degree.")
    retval = []
    if not periodic:
        retval += [start for _ in range(p - continuityDrop)]
    h = (end - start) / float(N)
    for i in range(0, N + 1):
        retval += [start + float(i) * h for _ in range(continuityDrop + 1)]
    if not periodic:
        retval += [end for _ in range(p - continuityDrop)]
    return retval


class BSpline1(object):
    """
    Scalar univariate B-spline (knot bookkeeping of tIGAr/BSplines.py:164-351).
    Evaluation methods run on the GPU.
    """

    def __init__(self, p, knots):
        self.p = int(p)
        self.knots = numpy.array(knots, dtype=numpy.float64)
        self.computeNel()
        # unique knots and multiplicities (needed for the FE node grid)
        newKnot = numpy.ones(len(self.knots), dtype=bool)
        for i in range(1, len(self.knots)):
            newKnot[i] = not near(self.knots[i], self.knots[i - 1], eps=KNOT_NEAR_EPS)
        # first knot of each run, exactly as the reference stores it
        self.uniqueKnots = self.knots[newKnot].copy()
        starts = numpy.flatnonzero(newKnot)
        self.multiplicities = numpy.diff(numpy.append(starts, len(self.knots))).astype(INDEX_TYPE)
        self.ncp = self.computeNcp()
        self.nGhost = self.p + 1
        self.ghostKnots = self.computeGhostKnots()

    def computeNel(self):
        """Number of non-degenerate knot spans (tIGAr/BSplines.py:236-244)."""
        self.nel = 0
        for i in range(1, len(self.knots)):
            if not near(self.knots[i], self.knots[i - 1], eps=KNOT_NEAR_EPS):
                self.nel += 1

    def getKnot(self, i):
        """Knot with a possibly out-of-range index: periodic ghost continuation
        (tIGAr/BSplines.py:246-260)."""
        n = len(self.knots)
        if i < 0:
            ii = n - int(self.multiplicities[-1]) + i
            return self.knots[0] - (self.knots[-1] - self.knots[ii])
        elif i >= n:
            ii = i - n + int(self.multiplicities[0])
            return self.knots[-1] + (self.knots[ii] - self.knots[0])
        return self.knots[i]

    def computeGhostKnots(self):
        return numpy.array([self.getKnot(i) for i in
                            range(-self.nGhost, len(self.knots) + self.nGhost)])

    def normalizeKnotVector(self):
        L = self.knots[-1] - self.knots[0]
        self.knots = (self.knots - self.knots[0]) / L
        self.uniqueKnots = (self.uniqueKnots - self.uniqueKnots[0]) / L
        self.ghostKnots = self.computeGhostKnots()

    def isDiscontinuous(self):
        return bool(numpy.any(self.multiplicities[1:-1] > self.p))

    def greville(self, i):
        """Greville parameter of the i-th control point (tIGAr/BSplines.py:262-271)."""
        retval = 0.0
        for j in range(i, i + self.p):
            retval += self.getKnot(j + 1)
        retval /= float(self.p)
        return retval

    def grevilleAll(self):
        """``greville(i)`` for every control point: the same additions in the same order (knots i+1 ... i+p, then / p), each
        knot looked up once instead of p times"""
        n, p = self.getNcp(), self.p
        kn = numpy.array([self.getKnot(j) for j in range(1, n + p)], dtype=numpy.float64)
        g = numpy.zeros(n)
        for q in range(p):
            g = g + kn[q:q + n]
        return g / float(p)

    def computeNcp(self):
        return len(self.knots) - int(self.multiplicities[0])

    def getNcp(self):
        return self.ncp

    # ---- evaluation: device twin of getKnotSpan / getNodes / basisFuncs -------------
    def evalBatch(self, us):
        """(spans, nodes[n,p+1], values[n,p+1]) for an array of parameters."""
        return _dev.eval_basis_1d(self, numpy.atleast_1d(numpy.asarray(us, dtype=numpy.float64)))

    def getKnotSpan(self, u):
        return int(self.evalBatch([u])[0][0])

    def getNodes(self, u):
        return [int(i) for i in self.evalBatch([u])[1][0]]

    def basisFuncs(self, knotSpan, u):
        """The p+1 basis values at ``u`` (tIGAr/BSplines.py:321-351).  The device twin derives the span
        from ``u`` exactly as ``getKnotSpan`` does (the reference's callers pass that value); any other
        span is refused instead of being silently ignored."""
        span, _, val = self.evalBatch([u])
        if knotSpan is not None and int(knotSpan) != int(span[0]):
            raise ValueError("basisFuncs: knotSpan %d is not getKnotSpan(u) = %d" % (int(knotSpan), int(span[0])))
        return val[0]

    # ---- FE node grid along this direction ------------------------------------------
    def feVertices(self, refine=0):
        """vertices of the extraction mesh in this direction: the unique knots, every span halved ``refine`` times
        (``refine(mesh)`` of tIGAr/BSplines.py:566-568 bisects every edge: midpoints exactly (a + b) / 2)"""
        uk = numpy.asarray(self.uniqueKnots, dtype=numpy.float64)
        for _ in range(int(refine)):
            mid = 0.5 * (uk[:-1] + uk[1:])
            out = numpy.empty(2 * len(uk) - 1)
            out[0::2] = uk
            out[1::2] = mid
            uk = out
        return uk

    def feNodes(self, degree, dg=False, refine=0):
        """
        Parametric coordinates of the Lagrange degree-``degree`` nodes on the knot mesh:
        equispaced reference points mapped affinely, x = x0*(1-t) + x1*t with t = j/degree,
        so vertex nodes lie exactly on the unique knots (the reference moves the dolfin
        mesh vertices onto ``uniqueKnots``, tIGAr/BSplines.py:527-536).  CG: nel*degree+1
        nodes; DG: nel*(degree+1).  ``refine``: on the mesh refined that many times.
        """
        uk = self.feVertices(refine)
        t = numpy.arange(degree + 1, dtype=numpy.float64) / float(degree)
        x = uk[:-1, None] * (1.0 - t[None, :]) + uk[1:, None] * t[None, :]
        x[:, 0] = uk[:-1]
        x[:, -1] = uk[1:]
        if dg:
            return x.reshape(-1).copy()
        return numpy.append(x[:, :-1].reshape(-1), uk[-1])


class _IndexList(list):
    """a list of dof indices that keeps the numpy array it came from (``.array``); every mutation through the list
    interface drops the array, so a consumer may trust ``.array`` while it is there (no pass over the entries)"""
    array = None

    def _dropped(name):                                    # noqa: N805
        def method(self, *a, **k):
            self.array = None
            return getattr(list, name)(self, *a, **k)
        method.__name__ = name
        return method
    for _n in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove",
               "reverse", "sort", "clear"):
        locals()[_n] = _dropped(_n)
    del _n, _dropped


def ij2dof(i, j, M):
    return j * M + i


def ijk2dof(i, j, k, M, N):
    return k * (M * N) + j * M + i


def dof2ij(dof, M):
    return (dof % M, dof // M)


def dof2ijk(dof, M, N):
    ij = dof % (M * N)
    return (ij % M, ij // M, dof // (M * N))


class BSpline(AbstractScalarBasis):
    """
    ``AbstractScalarBasis`` for a uni-, bi- or tri-variate B-spline
    (tIGAr/BSplines.py:374-649).
    """

    def __init__(self, degrees, kvecs, useRect=USE_RECT_ELEM_DEFAULT, overRefine=0):
        self.nvar = len(degrees)
        if self.nvar > 3 or self.nvar < 1:
            raise ValueError("Unsupported parametric dimension.")
        if overRefine and useRect:
            raise NotImplementedError("overRefine is only supported with simplicial elements "
                                      "(tIGAr/BSplines.py:393)")
        self.splines = [BSpline1(degrees[i], kvecs[i]) for i in range(self.nvar)]
        if not useRect and any(s.isDiscontinuous() for s in self.splines):
            raise NotImplementedError("simplicial extraction elements of a discontinuous spline space: the nodes of DG "
                                      "simplices are per triangle / tetrahedron, not a tensor lattice")
        self.useRect = useRect
        self.overRefine = overRefine
        self.ncp = self.computeNcp()
        self.nel = self.computeNel()

    def normalizeKnotVectors(self):
        for s in self.splines:
            s.normalizeKnotVector()

    def needsDG(self):
        return any(s.isDiscontinuous() for s in self.splines)

    def useRectangularElements(self):
        return self.useRect

    def getPrealloc(self):
        totalFuncs = 1
        for spline in self.splines:
            totalFuncs *= (spline.p + 1)
        return totalFuncs

    def getNodesAndEvalsBatch(self, X):
        """Vectorised ``getNodesAndEvals``: X is [n, nvar]; returns (cols[n,C], vals[n,C]) in
        the reference's entry order (i outer, then j, k innermost; value (Nu*Nv)*Nw)."""
        X = numpy.asarray(X, dtype=numpy.float64).reshape(-1, self.nvar)
        tabs = [self.splines[d].evalBatch(X[:, d]) for d in range(self.nvar)]
        n = X.shape[0]
        if self.nvar == 1:
            return tabs[0][1].astype(numpy.int64), tabs[0][2]
        M = self.splines[0].getNcp()
        if self.nvar == 2:
            cols = tabs[0][1][:, :, None].astype(numpy.int64) + M * tabs[1][1][:, None, :]
            vals = tabs[0][2][:, :, None] * tabs[1][2][:, None, :]
            return cols.reshape(n, -1), vals.reshape(n, -1)
        N = self.splines[1].getNcp()
        cols = (tabs[0][1][:, :, None, None].astype(numpy.int64) + M * tabs[1][1][:, None, :, None]
                + M * N * tabs[2][1][:, None, None, :])
        vals = (tabs[0][2][:, :, None, None] * tabs[1][2][:, None, :, None]) * tabs[2][2][:, None, None, :]
        return cols.reshape(n, -1), vals.reshape(n, -1)

    def getNodesAndEvals(self, xi):
        cols, vals = self.getNodesAndEvalsBatch(numpy.asarray(xi, dtype=numpy.float64).reshape(1, -1))
        return [[int(c), float(v)] for c, v in zip(cols[0], vals[0])]

    def generateMesh(self, comm=worldcomm, degree=None, dg=False):
        """The reference returns a dolfin mesh whose vertices are the unique knots
        (tIGAr/BSplines.py:505-569); here the "mesh" is the implicit tensor node grid."""
        deg = self.getDegree() if degree is None else degree
        # Simplicial elements (useRect=False, tIGAr/BSplines.py:520-523,543-549): dolfin's UnitSquareMesh / UnitCubeMesh
        # split every knot-span cell into 2 triangles / 6 tetrahedra, the extraction space is P_q on them with q = the
        # SUM of the degrees (getDegree), and ``overRefine`` bisects the edges that many times.  The P_q nodes of those
        # simplices are the points of the degree-q lattice of the (refined) cell, all of them: the node SET is the tensor
        # lattice below and M -- the basis evaluated at the nodes -- is the reference's up to dolfin's row numbering.
        # (FE objects assembled by ``forms`` use the Q_q cells of the same lattice: Q_q contains P_q, the spline space
        # lies in both, so M^T A M is the same Galerkin matrix of the spline space.)
        r = 0 if self.useRect else self.overRefine
        return TensorNodeGrid([s.feNodes(deg, dg, r) for s in self.splines],
                              [s.feVertices(r) for s in self.splines], deg, dg)

    def computeNcp(self):
        prod = 1
        for s in self.splines:
            prod *= s.getNcp()
        return prod

    def getNcp(self):
        return self.ncp

    def getDegree(self):
        deg = 0
        for s in self.splines:
            deg = max(deg, s.p) if self.useRect else deg + s.p
        return deg

    def computeNel(self):
        nel = 1
        for s in self.splines:
            nel *= s.nel
        return nel

    def getSideDofsArray(self, direction, side, nLayers=1):
        """``getSideDofs`` as one int64 array (same order)."""
        offsetSign = 1 - 2 * side
        ncps = [s.getNcp() for s in self.splines]
        ar = numpy.arange
        parts = []
        for absOffset in range(0, nLayers):
            i = (0 if side == 0 else ncps[direction] - 1) + absOffset * offsetSign
            if self.nvar == 1:
                parts.append(numpy.array([i], dtype=numpy.int64))
            elif self.nvar == 2:
                M, N = ncps
                parts.append(i + M * ar(N, dtype=numpy.int64) if direction == 0 else ar(M, dtype=numpy.int64) + M * i)
            else:
                M, N, O = ncps
                if direction == 0:          # (j outer, k inner: tIGAr/BSplines.py:627-630)
                    blk = i + M * ar(N, dtype=numpy.int64)[:, None] + M * N * ar(O, dtype=numpy.int64)[None, :]
                elif direction == 1:
                    blk = ar(M, dtype=numpy.int64)[:, None] + M * i + M * N * ar(O, dtype=numpy.int64)[None, :]
                else:
                    blk = ar(M, dtype=numpy.int64)[:, None] + M * ar(N, dtype=numpy.int64)[None, :] + M * N * i
                parts.append(blk.reshape(-1))
        return numpy.concatenate(parts) if parts else numpy.zeros(0, dtype=numpy.int64)

    def getSideDofs(self, direction, side, nLayers=1):
        """DoFs on a ``side`` (0 or 1) perpendicular to ``direction``; ``nLayers`` layers of
        control points.  Same ordering as tIGAr/BSplines.py:599-649 (corners repeat when
        several sides are concatenated).  A plain list, as in the reference; it remembers the array it
        was made from so that ``addZeroDofs`` need not convert 10^5 Python ints back."""
        arr = self.getSideDofsArray(direction, side, nLayers)
        out = _IndexList(arr.tolist())
        out.array = arr
        return out


class MultiBSpline(AbstractScalarBasis):
    """Several ``BSpline`` patches grouped into one scalar basis (tIGAr/BSplines.py:651-908): knot
    vectors are normalised to (0,1), patch ``i`` occupies x in [2i, 2i+1] of the parametric domain,
    DoFs of the patches are numbered one after the other.  (As in the reference there is no
    merging of control points between patches yet: the extraction operator is block diagonal.)"""

    PATCH_PITCH = 2.0           # parametric distance between the origins of neighbouring patches

    def __init__(self, splines):
        self.splines = splines
        if not len(splines):
            raise ValueError("MultiBSpline needs at least one patch")
        lead = splines[0]
        self.nvar, self.useRect, self.overRefine = lead.nvar, lead.useRect, lead.overRefine
        if self.nvar == 1:
            raise NotImplementedError("Univariate multipatch not yet supported.")   # (reference :745-747)
        self.nPatch = len(splines)
        for patch in splines:
            patch.normalizeKnotVectors()       # every patch on (0,1): patch index follows from x alone
        # DoFs of the patches are numbered one after the other: exclusive prefix sums of the patch sizes
        sizes = numpy.array([patch.getNcp() for patch in splines], dtype=numpy.int64)
        ends = numpy.cumsum(sizes)
        self.doffsets = [int(v) for v in (ends - sizes)]
        self.ncp = int(ends[-1])
        self.nel = self.computeNel()

    def computeNel(self):
        return int(sum(patch.nel for patch in self.splines))

    def computeNcp(self):
        return int(sum(patch.getNcp() for patch in self.splines))

    def getNcp(self):
        return self.ncp

    def needsDG(self):
        return False

    def useRectangularElements(self):
        return self.useRect

    def getPrealloc(self):
        return self.splines[0].getPrealloc()

    def getDegree(self):
        return max(patch.getDegree() for patch in self.splines)

    def patchFromCoordinates(self, xi):
        # patch i covers x in [2i, 2i+1]: round to the nearest integer, two units per patch
        return int(xi[0] + 0.5) // int(self.PATCH_PITCH)

    def globalDofIndex(self, localDofIndex, patchIndex):
        return localDofIndex + self.doffsets[patchIndex]

    def getPatchSideDofs(self, patch, direction, side, nLayers=1):
        """``BSpline.getSideDofs`` of patch ``patch``, in the global numbering (tIGAr/BSplines.py:898-908)."""
        local = self.splines[patch].getSideDofsArray(direction, side, nLayers)
        return (local + int(self.doffsets[patch])).tolist()

    def localParametricCoordinates(self, xi, patchIndex):
        shift = numpy.zeros(len(xi))
        shift[0] = self.PATCH_PITCH * float(patchIndex)
        return numpy.asarray(xi, dtype=numpy.float64) - shift

    def getNodesAndEvals(self, xi):
        ip = self.patchFromCoordinates(xi)
        off = self.doffsets[ip]
        local = self.splines[ip].getNodesAndEvals(self.localParametricCoordinates(xi, ip))
        return [[node + off, value] for node, value in local]

    def generateMesh(self, comm=worldcomm, degree=None, dg=False):
        """The reference writes a mesh of disconnected cells, 4 (8) vertices per element
        (tIGAr/BSplines.py:734-908); here: element-local node grids per patch."""
        from .common import MultiPatchNodeGrid
        deg = self.getDegree() if degree is None else degree
        return MultiPatchNodeGrid([s.generateMesh(comm=comm, degree=deg, dg=True) for s in self.splines])


class ExplicitBSplineControlMesh(AbstractControlMesh):
    """
    Control mesh of a B-spline with identical physical and parametric domains
    (tIGAr/BSplines.py:910-963): Greville abscissae, unit weights.
    """

    def __init__(self, degrees, kvecs, extraDim=0, useRect=USE_RECT_ELEM_DEFAULT, overRefine=0):
        self.scalarSpline = BSpline(degrees, kvecs, useRect=useRect, overRefine=overRefine)
        self.nvar = len(degrees)
        self.nsd = self.nvar + extraDim

    def getScalarSpline(self):
        return self.scalarSpline

    def getHomogeneousCoordinate(self, node, direction):
        if direction == self.nsd:
            return 1.0
        if direction < self.nvar:
            if self.nvar == 1:
                directionalIndex = node
            elif self.nvar == 2:
                directionalIndex = dof2ij(node, self.scalarSpline.splines[0].getNcp())[direction]
            else:
                M = self.scalarSpline.splines[0].getNcp()
                N = self.scalarSpline.splines[1].getNcp()
                directionalIndex = dof2ijk(node, M, N)[direction]
            return self.scalarSpline.splines[direction].greville(directionalIndex)
        return 0.0

    def getHomogeneousCoordinates(self):
        """All control points at once: array [ncp, nsd+1] (vectorised form of the per-node
        loop at tIGAr/common.py:373-375)."""
        sp_ = self.scalarSpline
        ncps = [s.getNcp() for s in sp_.splines]
        grev = [s.grevilleAll() for s in sp_.splines]
        P = numpy.zeros((sp_.getNcp(), self.nsd + 1))
        idx = numpy.arange(sp_.getNcp())
        stride = 1
        for d in range(self.nvar):
            P[:, d] = grev[d][(idx // stride) % ncps[d]]
            stride *= ncps[d]
        P[:, self.nsd] = 1.0
        return P

    def homogeneousCoordinateFactors(self, direction):
        """1-D factors of column ``direction`` of the control-point array: it is the tensor product of
        Greville abscissae along ``direction`` and ones elsewhere (weights: all ones; extra dimensions: zero)."""
        sp_ = self.scalarSpline
        ncps = [s.getNcp() for s in sp_.splines]
        if direction == self.nsd:
            return [numpy.ones(n) for n in ncps]
        if direction < self.nvar:
            facs = [numpy.ones(n) for n in ncps]
            s = sp_.splines[direction]
            facs[direction] = s.grevilleAll()
            return facs
        return [numpy.zeros(n) for n in ncps]

    def homogeneousCoordinateDeviceVector(self, direction):
        """Column ``direction`` of the control-point array as a device vector, expanded on the
        GPU from its 1-D factors."""
        return _dev.vec_tensor3(self.homogeneousCoordinateFactors(direction))

    def getNsd(self):
        return self.nsd
