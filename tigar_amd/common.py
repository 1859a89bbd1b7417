"""
The ``common`` module of tigar_amd: the reference's abstractions for generating extraction
data and using it in analysis (tIGAr/common.py), re-implemented MI355X-first.

What is kept: the ``AbstractExtractionGenerator`` / ``AbstractCoordinateChartSpline`` /
``AbstractMultiFieldSpline`` / ``EqualOrderSpline`` / ``FieldListSpline`` class tree, the
``AbstractScalarBasis`` / ``AbstractControlMesh`` plug-in interfaces, and
``ExtractedSpline`` with ``extractVector`` / ``extractMatrix`` / ``solveLinearSystem`` /
``setSolverOptions`` and the duck-typed ``linearSolver.solve(A, x, b)`` seam.

What changes: FEniCS objects are replaced by device-resident ones (``DeviceCSR`` for
``PETScMatrix``, ``DeviceVector`` for ``PETScVector``, ``TensorFunctionSpace`` for
``FunctionSpace`` on the implicit Q_p node grid).  FE assembly of forms is FEniCS's job
and is NOT part of this path; ``extractMatrix`` / ``extractVector`` accept any FE matrix /
vector on ``V`` (scipy.sparse / numpy inputs are uploaded).  Errors raise exceptions
instead of the reference's ``print("ERROR"); exit()``.
"""
import abc
import os
import sys
import numpy
from numpy import array, zeros, arange

from . import device as _dev
from .device import DeviceCSR, DeviceVector

# ---- module-level configuration, same names as tIGAr/common.py:35-84 ---------------------
mpisize = int(os.environ.get("WORLD_SIZE", "1"))
mpirank = int(os.environ.get("RANK", "0"))


class _Comm(object):
    """Stand-in for an MPI communicator handle: one process per GPU.  ``transport`` is the host-side message
    layer (``tigar_amd.launch.Transport``), ``device()`` the device communicator of the Krylov solve (RCCL over
    xGMI, or host-staged where ranks share a GPU); both are created on first use from the launcher's
    environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""

    def __init__(self, size, rank, transport=None, device_comm=None):
        self.size, self.rank = size, rank
        self._transport, self._device = transport, device_comm

    def transport(self):
        if self._transport is None:
            from .launch import SocketTransport, Transport
            self._transport = SocketTransport(self.rank, self.size) if self.size > 1 else Transport()
        return self._transport

    def device(self):
        if self.size > 1 and self._device is None:
            from .launch import device_comm
            self._device = device_comm(self.transport())
        return self._device

    def barrier(self):
        self.transport().barrier()


worldcomm = _Comm(mpisize, mpirank)
selfcomm = _Comm(1, 0)

INDEX_TYPE = 'int32'
DEFAULT_PREALLOC = 500
DEFAULT_DO_PERMUTATION = mpisize > 8
DEFAULT_BASIS_FUNC_IGNORE_EPS = 1e-15
EXTRACTION_DATA_FILE = "extraction-data.h5"
EXTRACTION_DATA_NPZ = "extraction-data.npz"       # what rounds 1-2 wrote in place of the HDF5 file (still read)
EXTRACTION_H5_PRIVATE = "/tigar_amd"               # group of extraction-data.h5 with this package's node-set description
EXTRACTION_INFO_FILE = "extraction-info.txt"
EXTRACTION_H5_MESH_NAME = "/mesh"


def EXTRACTION_H5_CONTROL_FUNC_NAME(dim):
    return "/control" + str(dim)


EXTRACTION_ZERO_DOFS_FILE = "zero-dofs.dat"
EXTRACTION_MAT_FILE = "extraction-mat.dat"
EXTRACTION_MAT_FILE_CTRL = "extraction-mat-ctrl.dat"
USE_DG_DEFAULT = True
USE_RECT_ELEM_DEFAULT = True
# The reference's FORM_MT=False trades speed for memory; on the GPU the explicit transpose is
# what makes M^T b and M^T A M scatter-free and deterministic, so it is always formed.
FORM_MT = True

DOLFIN_EPS = 3.0e-16


def generateMeshXMLFileName(comm):
    """name of the scratch mesh file of a communicator / rank pair (tIGAr/common.py:88-93; the product writes no such
    file -- the node grids are never serialised as XML -- but scripts that clean up after the reference call this)"""
    import hashlib
    s = repr(comm) + repr(getattr(comm, "rank", 0))
    return "mesh-" + str(hashlib.md5(s.encode("utf-8")).hexdigest()) + ".xml"


def near(a, b, eps=DOLFIN_EPS):
    """dolfin.near: |a-b| <= eps."""
    return abs(a - b) <= eps


# ---- FE-side stand-ins ------------------------------------------------------------------
class TensorNodeGrid(object):
    """The "mesh" of the extraction: a tensor-product grid of Lagrange nodes on the knot
    mesh (replaces the dolfin mesh of tIGAr/BSplines.py:505-569)."""

    def __init__(self, axes, vertices, degree, dg=False):
        self.axes = [numpy.ascontiguousarray(a, dtype=numpy.float64) for a in axes]
        self.vertices = vertices
        self.degree = degree
        self.dg = dg

    def dim(self):
        return len(self.axes)

    def shape(self):
        return [len(a) for a in self.axes]

    def num_nodes(self):
        n = 1
        for a in self.axes:
            n *= len(a)
        return n

    def coordinates(self):
        """[num_nodes, dim] parametric node coordinates, direction 0 fastest."""
        grids = numpy.meshgrid(*self.axes, indexing="ij")
        return numpy.stack([g.ravel(order="F") for g in grids], axis=1)


class MultiPatchNodeGrid(object):
    """Node set of a ``MultiBSpline``: the patches' node grids side by side, patch ``i`` shifted
    by 2*i along direction 0 (the reference's multi-patch mesh of disconnected elements,
    tIGAr/BSplines.py:734-860; each patch keeps element-local nodes)."""

    def __init__(self, patches):
        self.patches = list(patches)
        self.degree = self.patches[0].degree
        self.dg = True

    def dim(self):
        return self.patches[0].dim()

    def num_nodes(self):
        return sum(g.num_nodes() for g in self.patches)

    def coordinates(self):
        out = []
        for i, g in enumerate(self.patches):
            X = g.coordinates().copy()
            X[:, 0] += 2.0 * float(i)
            out.append(X)
        return numpy.vstack(out)


def _grid_to_arrays(g, key, data):
    """node grid -> entries of the ``extraction-data.npz`` archive under the prefix ``key`` (written by
    ``writeExtraction``, read back by ``_grid_from_arrays``)"""
    if isinstance(g, TensorNodeGrid):
        data[key + "_kind"] = numpy.array("tensor")
        data[key + "_degree"] = numpy.int64(g.degree)
        data[key + "_dg"] = numpy.int64(1 if g.dg else 0)
        data[key + "_dim"] = numpy.int64(g.dim())
        for k in range(g.dim()):
            data["%s_axis%d" % (key, k)] = numpy.asarray(g.axes[k])
            data["%s_vert%d" % (key, k)] = numpy.asarray(g.vertices[k], dtype=numpy.float64)
    elif isinstance(g, MultiPatchNodeGrid):
        data[key + "_kind"] = numpy.array("multipatch")
        data[key + "_npatches"] = numpy.int64(len(g.patches))
        for i, gp in enumerate(g.patches):
            _grid_to_arrays(gp, "%s_patch%d" % (key, i), data)
    elif type(g).__name__ == "BezierElementNodeGrid":
        data[key + "_kind"] = numpy.array("bezier")
        data[key + "_nel"] = numpy.int64(g.nel)
        data[key + "_degree"] = numpy.int64(g.degree)
    else:
        raise NotImplementedError("writeExtraction: node grid of type %s has no archive form" % type(g).__name__)


def _grid_from_arrays(data, key):
    kind = str(data[key + "_kind"]) if key + "_kind" in data else "tensor"       # (archives of round 1: tensor grids only)
    if kind == "tensor":
        dim = int(data[key + "_dim"])
        return TensorNodeGrid([data["%s_axis%d" % (key, k)] for k in range(dim)],
                              [data["%s_vert%d" % (key, k)] for k in range(dim)],
                              int(data[key + "_degree"]), bool(int(data[key + "_dg"])))
    if kind == "multipatch":
        return MultiPatchNodeGrid([_grid_from_arrays(data, "%s_patch%d" % (key, i))
                                   for i in range(int(data[key + "_npatches"]))])
    if kind == "bezier":
        from .RhinoTSplines import BezierElementNodeGrid
        return BezierElementNodeGrid(int(data[key + "_nel"]), int(data[key + "_degree"]))
    raise ValueError("extraction-data.npz: unknown node grid kind %r" % kind)


_CELL_NAMES = {1: "interval", 2: "quadrilateral", 3: "hexahedron"}


def _knot_mesh_arrays(g):
    """(coordinates [nv, d], topology [ncells, 2^d]) of the knot mesh under a tensor node grid: vertices and cells with
    direction 0 fastest, the vertices of a cell in lexicographic order -- the content of ``/mesh`` in extraction-data.h5
    (tIGAr/common.py:463).  The GROUP layout is dolfin's; whether dolfin's own quadrilateral / hexahedral meshes order
    the vertices of a cell this way has not been checked against dolfin (absent here), and nothing in this package
    depends on it: the node sets are rebuilt from the group ``/tigar_amd``."""
    d = g.dim()
    verts = [numpy.asarray(v, dtype=numpy.float64) for v in g.vertices]
    nv = [len(v) for v in verts]
    grids = numpy.meshgrid(*verts, indexing="ij")
    X = numpy.stack([q.ravel(order="F") for q in grids], axis=1)
    stride = numpy.cumprod([1] + nv[:-1])
    cells = numpy.meshgrid(*[numpy.arange(n - 1, dtype=numpy.int64) for n in nv], indexing="ij")
    base = sum(c.ravel(order="F") * int(stride[k]) for k, c in enumerate(cells))
    corners = numpy.zeros(1, dtype=numpy.int64)
    for k in range(d):                                 # lexicographic corners, direction 0 fastest
        corners = numpy.concatenate([corners, corners + int(stride[k])])
    return X, base[:, None] + corners[None, :]


def _kron_sum_is_symmetric(factors):
    """True when sum_t kron(F_t[d-1], ..., F_t[0]) is symmetric because of its 1-D factors: every term is a product of
    symmetric matrices, or its transpose (factor by factor) is another term of the sum (C^T x C + C x C^T).  1-D matrices:
    a few thousand entries, compared on the host."""
    if not factors:
        return False
    try:
        import scipy.sparse as _sp
        terms = [[_sp.csr_matrix(F) for F in t] for t in factors]

        def same(A, B):
            if A.shape != B.shape:
                return False
            D = abs(A - B)
            return (D.max() if D.nnz else 0.0) <= 1e-13 * max(abs(A).max() if A.nnz else 0.0, 1e-300)
        left = list(range(len(terms)))
        while left:
            t = left.pop(0)
            if all(same(F, F.T.tocsr()) for F in terms[t]):
                continue
            mate = next((u for u in left if len(terms[u]) == len(terms[t]) and
                         all(same(G, F.T.tocsr()) for F, G in zip(terms[t], terms[u]))), None)
            if mate is None:
                return False
            left.remove(mate)
        return True
    except Exception:
        return False


def _cell_dofs_arrays(g):
    """[ncells, (p+1)^d] node numbers of every knot-mesh cell, in this package's numbering (direction 0 fastest; a dolfin
    ``FunctionSpace`` on the same mesh numbers them its own way, which cannot be reproduced without dolfin)."""
    d, p = g.dim(), int(g.degree)
    nn = g.shape()
    step = p + 1 if g.dg else p
    nel = [(n // (p + 1)) if g.dg else ((n - 1) // max(p, 1)) for n in nn] if p > 0 else [n for n in nn]
    stride = numpy.cumprod([1] + nn[:-1])
    cells = numpy.meshgrid(*[numpy.arange(n, dtype=numpy.int64) for n in nel], indexing="ij")
    base = sum(c.ravel(order="F") * int(step * stride[k]) for k, c in enumerate(cells))
    loc = numpy.zeros(1, dtype=numpy.int64)
    for k in range(d):                                 # local nodes, direction 0 fastest
        loc = (loc[None, :] + (numpy.arange(p + 1, dtype=numpy.int64) * int(stride[k]))[:, None]).ravel()
    return base[:, None] + loc[None, :]


class _H5Archive(object):
    """the private group of extraction-data.h5 read like the npz archive of rounds 1-2 (arrays: datasets; strings and
    scalars: attributes of the group)"""

    def __init__(self, f):
        self.f = f

    def __contains__(self, key):
        return self.f.exists(EXTRACTION_H5_PRIVATE + "/" + key) or self.f.has_attr(EXTRACTION_H5_PRIVATE, key)

    def __getitem__(self, key):
        if self.f.exists(EXTRACTION_H5_PRIVATE + "/" + key):
            return self.f.read_dataset(EXTRACTION_H5_PRIVATE + "/" + key)
        if self.f.has_attr(EXTRACTION_H5_PRIVATE, key):
            return self.f.read_attr(EXTRACTION_H5_PRIVATE, key)
        raise KeyError(key)


class TensorFunctionSpace(object):
    """Stand-in for dolfin ``FunctionSpace``: ``nfields`` scalar Lagrange (or DG) fields
    on node grids that share one knot mesh.  Dofs are field-major, nodes lexicographic
    (dolfin's own numbering is unobservable downstream: K, M^T b and M U are invariant
    under FE row permutations -- SURVEY.md section 7 hard part 3)."""

    def __init__(self, grids, element):
        self.grids = list(grids)
        self.element = element

    def num_sub_spaces(self):
        return len(self.grids) if len(self.grids) > 1 else 0

    def dim(self):
        return sum(g.num_nodes() for g in self.grids)

    def field_offset(self, field):
        return sum(g.num_nodes() for g in self.grids[:field])

    def tabulate_dof_coordinates(self):
        return numpy.vstack([g.coordinates() for g in self.grids])


class Function(object):
    """Stand-in for dolfin ``Function``: FE coefficients in HBM."""

    def __init__(self, V, local_range=None, vector=None):
        self.V = V
        # with several ranks a function holds the FE rows [r0, r1) its rank owns (set by the solve)
        self.local_range = local_range
        # (several fields with several ranks: one range per field, a list of (r0, r1))
        if local_range is not None and len(local_range) and isinstance(local_range[0], (tuple, list)):
            nloc = sum(int(b) - int(a) for a, b in local_range)
        else:
            nloc = V.dim() if local_range is None else local_range[1] - local_range[0]
        self._vec = vector if vector is not None else DeviceVector(nloc)
        self._ghoster = None          # set by the spline that drives a distributed solve (ExtractedSpline.ghostedVector)

    def vector(self):
        return self._vec

    def ghosted(self):
        """FE coefficients as a vector of the full length of V, valid on every row the forms of THIS rank read: its own
        rows plus the ghost rows held by the z-neighbours (dolfin's ghost update of a distributed Function [ext]; the
        reference's forms read the ghosted vector after ``u.assign``, tIGAr/common.py:1343).  One rank: the vector itself.
        The exchange happens ONCE per assembly: the spline that drives the solve stamps its assemblies
        (``ExtractedSpline._ghost_epoch``) and the ghosted vector is kept until the stamp changes -- the forms ask for it once
        per row block, and the number of row blocks differs from rank to rank (the exchanges of z-neighbours
        would no longer pair up)."""
        if self.local_range is None:
            return self._vec
        if self._ghoster is None:
            raise RuntimeError("a rank-local Function can only be read by forms after the spline that owns it has been "
                               "asked for a ghost update (ExtractedSpline.ghostedVector / solveNonlinearVariationalProblem)")
        owner = getattr(self._ghoster, "__self__", None)
        epoch = getattr(owner, "_ghost_epoch", None)
        cached = self.__dict__.get("_ghost_cache")
        if epoch is not None and cached is not None and cached[0] == epoch and cached[1] is self._vec:
            return cached[2]
        g = self._ghoster(self)
        if epoch is not None:
            self.__dict__["_ghost_cache"] = (epoch, self._vec, g)
        return g

    def invalidate_ghosts(self):
        """Forget the ghost rows fetched for the current assembly: call after changing the coefficients in place outside the
        drivers (``vector().axpy`` / ``set_local``) when a form, functional or error norm reads the function next --
        ``solveLinearSystem`` and the Newton update do it themselves."""
        self.__dict__.pop("_ghost_cache", None)

    def function_space(self):
        return self.V


def _as_device_vector(b):
    if isinstance(b, DeviceVector):
        return b
    if hasattr(b, "vector") and callable(b.vector):
        return b.vector()
    return DeviceVector(data=numpy.asarray(b, dtype=numpy.float64))


def _as_device_csr(A):
    if isinstance(A, DeviceCSR):
        return A
    return DeviceCSR.from_scipy(A)


def multTranspose(M, b):
    """Returns ``M^T*b`` (tIGAr/common.py:97-109)."""
    return M.mult_transpose(_as_device_vector(b))


def generateIdentityPermutation(ownRange, comm=worldcomm):
    """Index array of the ownership range (tIGAr/common.py:114-128)."""
    return arange(ownRange[0], ownRange[1], dtype=INDEX_TYPE)


# ---- extraction generators -----------------------------------------------------------------
class AbstractExtractionGenerator(object):
    """
    Minimal set of functions needed to write extraction operators for a spline
    (tIGAr/common.py:130-502).
    """

    __metaclass__ = abc.ABCMeta

    def __init__(self, comm, *args):
        if not isinstance(comm, _Comm):
            args = (comm,) + args
            self.comm = worldcomm
        else:
            self.comm = comm
        self.customSetup(args)
        self.genericSetup()

    def getComm(self):
        return self.comm

    def useDG(self):
        return USE_DG_DEFAULT

    def extractionElement(self):
        return "DG" if self.useDG() else "Lagrange"

    @abc.abstractmethod
    def customSetup(self, args):
        return

    @abc.abstractmethod
    def getNFields(self):
        return

    @abc.abstractmethod
    def getHomogeneousCoordinate(self, node, direction):
        return

    @abc.abstractmethod
    def generateMesh(self):
        return

    @abc.abstractmethod
    def getDegree(self, field):
        return

    @abc.abstractmethod
    def getNcp(self, field):
        return

    @abc.abstractmethod
    def getNsd(self):
        return

    def globalDof(self, field, localDof):
        retval = localDof
        for i in range(0, field):
            retval += self.getNcp(i)
        return retval

    def generatePermutation(self):
        total = sum(self.getNcp(i) for i in range(self.getNFields()))
        return generateIdentityPermutation((0, total), self.comm)

    # ``zeroDofs`` is a plain Python list in the reference (tIGAr/common.py:254-282; user code appends to it).  Large
    # index sets (6 faces x 259^2 at cfg3) are kept as numpy chunks behind it: the list is only materialised when
    # somebody reads the attribute, the path itself takes ``zeroDofsArray()``.
    @property
    def zeroDofs(self):
        chunks = self.__dict__.get("_zero_chunks", [])
        if chunks:
            self.__dict__["_zero_list"] = self.__dict__.get("_zero_list", []) + numpy.concatenate(chunks).tolist()
            self.__dict__["_zero_chunks"] = []
        return self.__dict__.setdefault("_zero_list", [])

    @zeroDofs.setter
    def zeroDofs(self, value):
        # (``self.zeroDofs += [dof]`` hands back the very list the getter returned: keep it, do not copy it per append)
        if value is not self.__dict__.get("_zero_list"):
            self.__dict__["_zero_list"] = list(value)
        self.__dict__["_zero_chunks"] = []

    def zeroDofsArray(self):
        """all zero dofs so far, in insertion order (duplicates kept), as one int64 array"""
        parts = [numpy.asarray(self.__dict__.get("_zero_list", []), dtype=numpy.int64)] + \
            self.__dict__.get("_zero_chunks", [])
        return numpy.concatenate(parts) if len(parts) > 1 else parts[0]

    @staticmethod
    def _index_array(dofs):
        # a list made by getSideDofs carries its numpy twin until the list is edited (every mutating method of the list
        # drops it): no pass over 67 000 Python integers per face at 256^3
        src = getattr(dofs, "array", None)
        if src is not None and type(dofs).__name__ == "_IndexList" and len(src) == len(dofs):
            return src
        return numpy.asarray(dofs, dtype=numpy.int64).reshape(-1)

    def addZeroDofsGlobal(self, newDofs):
        self.__dict__.setdefault("_zero_chunks", []).append(numpy.array(self._index_array(newDofs), dtype=numpy.int64))

    def addZeroDofs(self, field, newDofs):
        off = self.globalDof(field, 0)
        self.__dict__.setdefault("_zero_chunks", []).append(self._index_array(newDofs) + off)

    def getPrealloc(self, control):
        return DEFAULT_PREALLOC

    def getIgnoreEps(self):
        return DEFAULT_BASIS_FUNC_IGNORE_EPS

    @abc.abstractmethod
    def generateM_control(self):
        return

    @abc.abstractmethod
    def generateM(self):
        return

    def _homogeneousCoordinateArray(self):
        """P[I, i] = getHomogeneousCoordinate(I, i) (loop of tIGAr/common.py:373-375),
        vectorised when the control mesh offers it."""
        cm = getattr(self, "getControlMesh", None)
        if cm is not None and hasattr(self.getControlMesh(), "getHomogeneousCoordinates"):
            return numpy.asarray(self.getControlMesh().getHomogeneousCoordinates(), dtype=numpy.float64)
        ncp = self.getNcp(-1)
        P = numpy.empty((ncp, self.nsd + 1))
        for I in range(ncp):
            for i in range(self.nsd + 1):
                P[I, i] = self.getHomogeneousCoordinate(I, i)
        return P

    def genericSetup(self):
        """Common setup (tIGAr/common.py:321-383): node grids in place of the FE mesh and
        function spaces, M_control, M, and the FE-nodal control functions
        cpFuncs[i] = M_control * P[:, i]."""
        self.mesh = self.generateMesh()
        self.nsd = self.getNsd()
        elem = self.extractionElement()
        dg = (elem == "DG")
        self.V_control = TensorFunctionSpace([self._fieldGrid(-1, dg)], elem)
        self.V = TensorFunctionSpace([self._fieldGrid(i, dg) for i in range(self.getNFields())], elem)
        self.M_control = self.generateM_control()
        self.M = self.generateM()
        self.MT = self.generateMT()
        self.M._T = self.MT           # M.mult_transpose() goes through the explicit transpose
        # Kronecker form of M (single tensor B-spline field, filter dropped only exact zeros):
        # enables the sum-factorised M^T A M of tigar_amd/kronptap.py
        self._kron = None
        self._kron_scalar = None           # several fields on ONE tensor basis (EqualOrderSpline(nFields > 1)): its tables
        self._kron_fields = None           # several fields on different tensor bases: one set of tables per field
        if getattr(self.M, "is_implicit", False) and hasattr(self.M, "kxs"):
            self._kron_fields = self.M.kxs       # diag(M_0, ..., M_{nF-1}) on different bases, never stored
        elif getattr(self.M, "is_implicit", False) and getattr(self.M, "nfields", 1) > 1:
            self._kron_scalar = self.M.kx        # diag(M_s, ..., M_s), never stored
        elif getattr(self.M, "is_implicit", False):
            self._kron = self.M.kx
        elif self.getNFields() == 1 and (0, 0) in self._fast_blocks or (self.M is self.M_control
                                                                          and (-1, 0) in self._fast_blocks):
            basis, grid = self._fast_blocks.get((0, 0), self._fast_blocks.get((-1, 0)))
            kx = self._kron_tables(basis, grid)
            if kx is not None and kx.is_exact_for(self.M.nnz, self.getIgnoreEps()):
                self._kron = kx
        elif self.getNFields() > 1 and not getattr(self.M, "is_implicit", False):
            # every field block built by the tensor kernels from the same basis on the same node grid, and each block
            # exactly the Kronecker product of its 1-D factors (entry count checked): M = diag(M_s, ..., M_s)
            keys = []
            offset = 0
            for f in range(self.getNFields()):
                keys.append((f, offset))
                offset += self.getNcp(f)
            if all(k in self._fast_blocks for k in keys):
                b0, g0 = self._fast_blocks[keys[0]]
                same = all(self._fast_blocks[k][0] is b0 and self._fast_blocks[k][1].axes is not None and
                           all(numpy.array_equal(a, b) for a, b in zip(self._fast_blocks[k][1].axes, g0.axes))
                           for k in keys)
                kx = self._kron_tables(b0, g0) if same else None
                if kx is not None and self.M.nnz == self.getNFields() * kx.nnz_product and \
                        kx.products_stay_above(self.getIgnoreEps()):
                    self._kron_scalar = kx
                if self._kron_scalar is None:
                    # fields on DIFFERENT tensor bases (FieldListSpline, the components of a compatible spline:
                    # tIGAr/common.py:1949-1970, tIGAr/compatibleSplines.py:21-101): M = diag(M_0, ..., M_{nF-1}), every
                    # block exactly the Kronecker product of its own 1-D factors (entry count checked)
                    kxs = [self._kron_tables(*self._fast_blocks[k]) for k in keys]
                    if all(kx is not None and kx.products_stay_above(self.getIgnoreEps()) for kx in kxs) and \
                            self.M.nnz == sum(kx.nnz_product for kx in kxs):
                        self._kron_fields = kxs
        self.cpFuncs = []
        cm = self.getControlMesh() if hasattr(self, "getControlMesh") else None
        P = None
        self._slab_engine = None
        if self.comm.size > 1 and getattr(self.M_control, "is_implicit", False):
            # several ranks: every rank evaluates the control functions on the FE rows it owns only
            # (the reference's cpFuncs are distributed dolfin Functions, tIGAr/common.py:367-380)
            from .dist import SlabHotPath
            kx = self.M_control.kx
            self._slab_engine = SlabHotPath(kx.basis, kx.grid, self.comm.rank, self.comm.size, self.comm.device(),
                                            sub_planes="auto", eps=self.M_control.eps, kx=kx)
        # M_control a Kronecker product (implicit, or stored and checked entry count against the product of the
        # 1-D counts) and the control net a tensor product of 1-D factors: so is every control function
        kx_c = None
        if cm is not None and hasattr(cm, "homogeneousCoordinateFactors"):
            if getattr(self.M_control, "is_implicit", False):
                kx_c = self.M_control.kx
            elif self._kron is not None and self.M is self.M_control:
                kx_c = self._kron
        separable = kx_c is not None
        self._cp_source = {"kx": kx_c if separable else (self.M_control.kx if getattr(self.M_control, "is_implicit", False) else None),
                           "separable": separable, "P": []}
        for i in range(self.nsd + 1):
            if separable:
                # (M_z g_z) (x) (M_y g_y) (x) (M_x g_x) -- one write pass over the FE rows this rank owns
                kx = kx_c
                facs = cm.homogeneousCoordinateFactors(i)
                fe1d = [kx.M1[k] @ numpy.asarray(facs[k], dtype=numpy.float64) for k in range(kx.d)]
                rng = self._slab_engine.mine["u_rows"] if self._slab_engine is not None else None
                self.cpFuncs += [Function(self.V_control, rng,
                                          vector=_dev.vec_tensor3(fe1d, 1.0, rng[0] if rng else None, rng[1] if rng else None))]
                continue
            if cm is not None and hasattr(cm, "homogeneousCoordinateDeviceVector"):
                Pi = cm.homogeneousCoordinateDeviceVector(i)      # built in HBM from 1-D factors
            else:
                if P is None:
                    P = self._homogeneousCoordinateArray()
                Pi = DeviceVector(data=P[:, i])
            self._cp_source["P"].append(Pi)
            if self._slab_engine is not None:
                f = Function(self.V_control, self._slab_engine.mine["u_rows"])
                f._vec = self._slab_engine._prolong_tensor(Pi, 0)
            else:
                f = Function(self.V_control)
                self.M_control.mult(Pi, f.vector())               # stays in HBM
            self.cpFuncs += [f]
        self.zeroDofs = []

    def controlFunctionWindow(self, fa, fb):
        """The nsd+1 control functions cpFuncs[i] = M_control P[:, i] (tIGAr/common.py:367-380) on the FE node planes
        [fa, fb) of the last parametric direction, as DeviceVectors -- what the forms of a rank read when they assemble a
        block of FE rows on a mapped patch.  Evaluated from the control net (replicated on every rank) through the rows
        of M_control that belong to the window: no rank needs another rank's part of the control functions.  The last
        two windows are kept (the matrix and the vector form of one sub-slab ask for the same one)."""
        key = (int(fa), int(fb))
        cache = self.__dict__.setdefault("_cp_windows", [])
        for k_, v_ in cache:
            if k_ == key:
                return v_
        src = self.__dict__.get("_cp_source") or {}
        kx = src.get("kx")
        if kx is None:
            raise NotImplementedError("control functions on a window of FE planes: tensor-product control mesh")
        pf = int(numpy.prod(kx.nfe[:-1], dtype=numpy.int64)) if kx.d > 1 else 1
        out = []
        if src["separable"]:
            cm = self.getControlMesh()
            for i in range(self.nsd + 1):
                facs = cm.homogeneousCoordinateFactors(i)
                fe1d = [kx.M1[k] @ numpy.asarray(facs[k], dtype=numpy.float64) for k in range(kx.d)]
                out.append(_dev.vec_tensor3(fe1d, 1.0, key[0] * pf, key[1] * pf))
        else:
            # rows [fa, fb) of the last 1-D factor name the dof planes [k_lo, k_hi): only those planes of the control net
            # go through the plane-local passes
            Mz = kx.M1[-1][key[0]:key[1]].tocsr()
            k_lo, k_hi = int(Mz.indices.min()), int(Mz.indices.max()) + 1
            pd = int(numpy.prod(kx.ncp[:-1], dtype=numpy.int64)) if kx.d > 1 else 1
            for Pi in src["P"]:
                x = DeviceVector((k_hi - k_lo) * pd)
                _dev.vec_copy_range(x, 0, Pi, k_lo * pd, (k_hi - k_lo) * pd)
                dims = list(kx.ncp[:-1]) + [k_hi - k_lo]
                t = x
                for k in range(kx.d - 1):
                    t = _dev.tensor_apply_1d(t, dims, k, kx.M1[k])
                    dims[k] = kx.nfe[k]
                out.append(_dev.tensor_apply_1d(t, dims, kx.d - 1, Mz, col_shift=k_lo))
        cache.append((key, out))
        del cache[:-2]
        return out

    def _fieldGrid(self, field, dg):
        """Node grid of one scalar field (degree getDegree(field) on the shared knot mesh)."""
        raise NotImplementedError

    def applyPermutation(self, nparts=None, fe_owner=None):
        """Renumbers the IGA dofs with ``generatePermutation`` (tIGAr/common.py:407-433): the columns of M move
        (MatPermute with the identity on the rows: new column j = old column perm[j]) and the zero dofs are renamed
        (the AO of :425-433: old dof a becomes the j with perm[j] = a).  With one part -- and for tensor patches in
        z-slabs, whose dof slabs already follow the FE slabs -- the permutation is the identity and nothing moves."""
        kw = {}
        if nparts is not None or fe_owner is not None:
            kw = {"nparts": nparts, "fe_owner": fe_owner}
        self.permutation = perm = numpy.asarray(self.generatePermutation(**kw), dtype=numpy.int64)
        if perm.size == 0 or numpy.array_equal(perm, numpy.arange(perm.size)):
            return
        new_of_old = numpy.empty(perm.size, dtype=numpy.int64)
        new_of_old[perm] = numpy.arange(perm.size)
        self.M = self.M.permute_columns(new_of_old)
        self.MT = self.M.transpose()
        self._kron = None                      # (M is no Kronecker product in the new numbering)
        self._kron_scalar = None
        self._kron_fields = None
        self._fast_blocks = {}
        self.zeroDofs = new_of_old[self.zeroDofsArray()].tolist()

    def writeExtraction(self, dirname, doPermutation=DEFAULT_DO_PERMUTATION, nparts=None):
        """Writes the extraction data to ``dirname`` with the reference's file names and formats
        (tIGAr/common.py:435-502): ``extraction-mat.dat`` / ``extraction-mat-ctrl.dat`` (PETSc binary
        Mat of M / M_control), ``zero-dofs.dat`` (PETSc binary IS), ``extraction-info.txt`` (nsd,
        element type, number of fields, then degree and ncp of the control field and of each field).
        ``extraction-data.h5`` is written through libhdf5 (``tigar_amd/h5io.py``) with the groups dolfin's ``HDF5File``
        writes (tIGAr/common.py:460-467): ``/mesh`` (``coordinates``, ``topology`` with its ``celltype``, ``cell_indices``;
        tensor node sets) and ``/control<i>`` (``vector_0``, ``cell_dofs``, ``x_cell_dofs``, ``cells``, the element's
        ``signature``) -- in THIS package's node numbering (dolfin's own cannot be reproduced without dolfin; the layout
        follows dolfin 2019's HDF5File and is not pinned against a dolfin-written file) -- plus the group ``/tigar_amd``
        with the node-set description ``initFromFilesystem`` rebuilds the function spaces from."""
        from . import petscio
        if doPermutation:
            self.applyPermutation(nparts=nparts)      # (nparts: the number of ranks that will read the directory)
        os.makedirs(dirname, exist_ok=True)
        petscio.write_mat(os.path.join(dirname, EXTRACTION_MAT_FILE), self.M.to_scipy())
        petscio.write_mat(os.path.join(dirname, EXTRACTION_MAT_FILE_CTRL), self.M_control.to_scipy())
        petscio.write_is(os.path.join(dirname, EXTRACTION_ZERO_DOFS_FILE), numpy.asarray(self.zeroDofs, dtype=INDEX_TYPE))
        fs = str(self.getNsd()) + "\n" + self.extractionElement() + "\n" + str(self.getNFields()) + "\n"
        for i in range(-1, self.getNFields()):
            fs += str(self.getDegree(i)) + "\n" + str(self.getNcp(i)) + "\n"
        with open(os.path.join(dirname, EXTRACTION_INFO_FILE), "w") as f:
            f.write(fs)
        data = {"nfields": numpy.int64(self.getNFields())}
        for name, V in (("control", self.V_control), ("fields", self.V)):
            data[name + "_ngrids"] = numpy.int64(len(V.grids))
            data[name + "_element"] = numpy.array(V.element)
            for gi, g in enumerate(V.grids):
                _grid_to_arrays(g, "%s_%d" % (name, gi), data)
        from . import h5io
        if not h5io.available():
            # no libhdf5 on this machine: the archive rounds 1-2 wrote (still read by ExtractedSpline(dirname))
            import warnings
            warnings.warn("libhdf5 not found (TIGAR_HDF5_LIB): writing %s instead of %s" % (EXTRACTION_DATA_NPZ, EXTRACTION_DATA_FILE))
            for i, fn in enumerate(self.cpFuncs):
                data["control%d" % i] = fn.vector().get_local()
            numpy.savez(os.path.join(dirname, EXTRACTION_DATA_NPZ), **data)
            return
        with h5io.H5File(os.path.join(dirname, EXTRACTION_DATA_FILE), "w") as f:
            g0 = self.V_control.grids[0]
            cell_dofs = None
            if isinstance(g0, TensorNodeGrid):
                X, topo = _knot_mesh_arrays(g0)
                f.create_group(EXTRACTION_H5_MESH_NAME)
                f.write_dataset(EXTRACTION_H5_MESH_NAME + "/coordinates", X)
                f.write_dataset(EXTRACTION_H5_MESH_NAME + "/topology", topo,
                                attrs={"celltype": _CELL_NAMES[g0.dim()], "partition": numpy.zeros(1, dtype=numpy.uint64)})
                f.write_dataset(EXTRACTION_H5_MESH_NAME + "/cell_indices", numpy.arange(topo.shape[0], dtype=numpy.int64))
                cell_dofs = _cell_dofs_arrays(g0)
                family = {1: ("Lagrange", "Discontinuous Lagrange"), 2: ("Q", "DQ"), 3: ("Q", "DQ")}[g0.dim()][1 if g0.dg else 0]
                signature = "FiniteElement('%s', %s, %d)" % (family, _CELL_NAMES[g0.dim()], int(g0.degree))
            for i, fn in enumerate(self.cpFuncs):
                name = EXTRACTION_H5_CONTROL_FUNC_NAME(i)
                f.create_group(name)
                f.write_dataset(name + "/vector_0", fn.vector().get_local(), attrs={"partition": numpy.zeros(1, dtype=numpy.uint64)})
                if cell_dofs is not None:
                    f.write_dataset(name + "/cell_dofs", cell_dofs.ravel())
                    f.write_dataset(name + "/x_cell_dofs", numpy.arange(cell_dofs.shape[0] + 1, dtype=numpy.int64) * cell_dofs.shape[1])
                    f.write_dataset(name + "/cells", numpy.arange(cell_dofs.shape[0], dtype=numpy.int64))
                    f.write_attr(name, "signature", signature)
            f.create_group(EXTRACTION_H5_PRIVATE)
            for k, v in data.items():
                v = numpy.asarray(v)
                if v.dtype.kind in "US":
                    f.write_attr(EXTRACTION_H5_PRIVATE, k, str(v))
                elif v.ndim == 0:
                    f.write_attr(EXTRACTION_H5_PRIVATE, k, v.astype(numpy.int64 if v.dtype.kind in "iub" else numpy.float64))
                else:
                    f.write_dataset(EXTRACTION_H5_PRIVATE + "/" + k, v.astype(numpy.int64 if v.dtype.kind in "iub" else numpy.float64))


class AbstractCoordinateChartSpline(AbstractExtractionGenerator):
    """Spline with a single parametric coordinate chart (tIGAr/common.py:1435-1669)."""

    @abc.abstractmethod
    def getNodesAndEvals(self, x, field):
        return

    def _generate_block(self, field, col_offset, ncols, grid, support_only=False):
        """Rows of the extraction matrix for one field: kernel path for tensor B-splines,
        host-loop triplet fallback for arbitrary AbstractScalarBasis plug-ins (seam b-2).
        ``support_only``: every function ``getNodesAndEvals`` returns is kept whatever its value (the pattern
        ``generatePermutation`` works on, tIGAr/common.py:1621-1629), always as a stored matrix."""
        basis = self.getScalarSpline(field) if hasattr(self, "getScalarSpline") else None
        eps = -1.0 if support_only else self.getIgnoreEps()
        from .BSplines import BSpline
        if isinstance(basis, BSpline) and type(basis).getNodesAndEvals is BSpline.getNodesAndEvals:
            if support_only:
                return _dev.extract_csr_tensor(basis.splines, grid.axes, col_offset, ncols, eps)
            self._fast_blocks[(field, col_offset)] = (basis, grid)
            if field == -1 and (self._single_shared_field() or self._fields_on_control_basis()):
                lazy = self._implicit_block(basis, grid, eps, several_fields=not self._single_shared_field())
                if lazy is not None:
                    return lazy
            # exact Kronecker form (every product of stored 1-D entries passes the filter, checked on the 1-D
            # tables; periodic directions included as long as a node names no function twice): pencil walk with
            # closed-form row starts, same entries bit for bit
            kx = self._kron_tables(basis, grid)
            if kx is not None and kx.products_stay_above(eps) and kx.columns_distinct() \
                    and os.environ.get("TIGAR_EXTRACT_KRON", "1") != "0":
                return _dev.kron3_csr(kx.M1, None, None, col_offset, ncols)
            return _dev.extract_csr_tensor(basis.splines, grid.axes, col_offset, ncols, eps)
        if hasattr(basis, "extractBlockOnDevice") and type(self).getNodesAndEvals is AbstractMultiFieldSpline.getNodesAndEvals:
            # bases that bring their own batched device evaluation (Rhino T-splines: csrc/tg_bezier.hip)
            return basis.extractBlockOnDevice(grid, col_offset, ncols, eps)
        from .BSplines import MultiBSpline
        if isinstance(basis, MultiBSpline) and isinstance(grid, MultiPatchNodeGrid):
            # patch by patch with the tensor kernel; rows stack, columns shift by the patch offsets
            blocks = [_dev.extract_csr_tensor(sp.splines, g.axes, col_offset + basis.doffsets[i], ncols, eps)
                      for i, (sp, g) in enumerate(zip(basis.splines, grid.patches))]
            return blocks[0] if len(blocks) == 1 else _dev.csr_vstack(blocks)
        # generic path: the reference's row loop (tIGAr/common.py:1554-1571)
        X = grid.coordinates()
        rows, cols, vals = [], [], []
        for I in range(X.shape[0]):
            for c, v in self.getNodesAndEvals(X[I], field):
                rows.append(I)
                cols.append(int(c) + col_offset)
                vals.append(float(v))
        return _dev.csr_from_triplets(X.shape[0], ncols, rows, cols, vals, eps)

    def _single_shared_field(self):
        """one unknown field on the control mesh's own basis: M is M_control"""
        return False

    def _fields_on_control_basis(self):
        """every unknown field is discretised with the control mesh's scalar basis (EqualOrderSpline): M is the block
        diagonal of M_control"""
        return False

    def _kron_tables(self, basis, grid):
        """1-D extraction tables of a tensor basis on its grid (``KronExtraction``), built once per (basis, grid)"""
        cache = self.__dict__.setdefault("_kx_cache", {})
        key = (id(basis), id(grid))
        if key not in cache:
            from .kronptap import KronExtraction
            # the fields of an equal-order spline (and its control field) sit on node grids of equal content: one set of
            # 1-D tables serves them all (each KronExtraction evaluates the basis at every node of every direction)
            twin = None
            for (kx2, b2, g2) in cache.values():
                if kx2 is not None and b2 is basis and g2.degree == grid.degree and bool(g2.dg) == bool(grid.dg) and \
                        len(g2.axes) == len(grid.axes) and all(numpy.array_equal(a, b) for a, b in zip(g2.axes, grid.axes)):
                    twin = kx2
                    break
            try:
                cache[key] = (twin if twin is not None else KronExtraction(basis, grid), basis, grid)   # (objects kept alive with the key)
            except Exception:
                cache[key] = (None, basis, grid)
        return cache[key][0]

    def _implicit_block(self, basis, grid, eps, several_fields=False):
        """``ImplicitExtraction`` in place of the CSR block when M (and M^T) would not fit beside the rest of
        the path -- 24 B per entry against a third of the free HBM -- or when the patch is spread over several
        ranks (no rank ever holds all rows); TIGAR_IMPLICIT_M=1/0 forces / forbids it.  Only offered when M
        is exactly the Kronecker product of its 1-D factors (the filter of tIGAr/common.py:1569 dropped
        nothing but exact zeros)."""
        from .kronptap import KronExtraction
        from .implicit import ImplicitExtraction
        env = os.environ.get("TIGAR_IMPLICIT_M")
        if env == "0":
            return None
        kx = self._kron_tables(basis, grid)
        if kx is None or not kx.products_stay_above(eps):
            return None
        if self.comm.size == 1:
            # the sub-slab engine behind an implicit M cuts the LAST direction into slabs and needs an open knot vector
            # there (tigar_amd/dist.py: ZSlabLayout); a patch that is periodic in that direction keeps a stored M on one
            # rank -- also when TIGAR_IMPLICIT_M=1 asks for the implicit one
            s_last = kx.basis.splines[-1]
            kn, pl = numpy.asarray(s_last.knots, dtype=float), int(s_last.p)
            if not (numpy.all(kn[:pl + 1] == kn[0]) and numpy.all(kn[-(pl + 1):] == kn[-1])):
                return None
        if env != "1" and self.comm.size == 1:
            free_b = _dev.mem_info()[0] + _dev.pool_stats()[0]      # idle blocks of the caching allocator count as free
            nf = self.getNFields() if several_fields else 1
            if several_fields or 24.0 * nf * kx.nnz_product + 16.0 * nf * grid.num_nodes() <= free_b / 3.0:
                return None                     # (several fields on one rank: stored, block by block as before)
        return ImplicitExtraction(kx, eps)

    def generateM_control(self):
        """Extraction matrix of the scalar space of the control functions
        (tIGAr/common.py:1460-1514)."""
        self._fast_blocks = {}
        return self._generate_block(-1, 0, self.getNcp(-1), self.V_control.grids[0])

    def generateMT(self):
        """Explicit transpose of ``M`` (the reference's optional ``MT``, FORM_MT at
        tIGAr/common.py:84,358-360).  For tensor B-spline fields it is written directly by the
        transposed extraction kernel; otherwise ``M`` is transposed on the device."""
        nf = self.getNFields()
        if getattr(self.M, "is_implicit", False):
            return self.M.transpose()
        def transposed_block(basis, grid, fe_offset, fe_total):
            kx = self._kron_tables(basis, grid)
            if kx is not None and kx.products_stay_above(self.getIgnoreEps()) and kx.columns_distinct() \
                    and os.environ.get("TIGAR_EXTRACT_KRON", "1") != "0":
                return _dev.kron3_csr(kx.M1T, None, None, fe_offset, fe_total)
            return _dev.extract_csr_tensor_t(basis.splines, grid.axes, fe_offset, fe_total, self.getIgnoreEps())
        if self.M is self.M_control and (-1, 0) in self._fast_blocks:
            basis, grid = self._fast_blocks[(-1, 0)]
            return transposed_block(basis, grid, 0, grid.num_nodes())
        blocks = []
        offset = 0
        fe_total = self.V.dim()
        for field in range(nf):
            key = (field, offset)
            if key not in self._fast_blocks:
                return self.M.transpose()
            basis, grid = self._fast_blocks[key]
            blocks.append(transposed_block(basis, grid, self.V.field_offset(field), fe_total))
            offset += self.getNcp(field)
        return blocks[0] if len(blocks) == 1 else _dev.csr_vstack(blocks)

    def generateM(self, support_only=False):
        """Extraction matrix of the mixed space of all unknown fields
        (tIGAr/common.py:1516-1578): one row block per field, column offset = sum of the
        previous fields' ncp."""
        totalDofs = sum(self.getNcp(i) for i in range(self.getNFields()))
        if self.getNFields() > 1 and not support_only and self._fields_on_control_basis() and \
                getattr(getattr(self, "M_control", None), "is_implicit", False):
            # several ranks (or TIGAR_IMPLICIT_M=1): no rank holds all rows -- the block diagonal of the implicit
            # scalar operator stands in for the matrix
            from .implicit import BlockImplicitExtraction
            return BlockImplicitExtraction(self.M_control.kx, self.getNFields(), self.M_control.eps)
        if self.getNFields() > 1 and not support_only and not self._fields_on_control_basis():
            lazy = self._implicit_field_list()
            if lazy is not None:
                return lazy
        blocks = []
        offset = 0
        for field in range(self.getNFields()):
            blocks.append(self._generate_block(field, offset, totalDofs, self.V.grids[field], support_only))
            offset += self.getNcp(field)
        return blocks[0] if len(blocks) == 1 else _dev.csr_vstack(blocks)

    def _implicit_field_list(self):
        """``FieldListImplicitExtraction`` for fields on different tensor B-spline bases over ONE node grid when the patch is
        spread over several ranks (no rank ever holds all rows) or TIGAR_IMPLICIT_M=1 asks for it; every field's operator
        must be exactly the Kronecker product of its 1-D factors (the filter of tIGAr/common.py:1569 dropped nothing but
        exact zeros) and the slab direction must carry open knot vectors.  None otherwise (stored blocks)."""
        from .BSplines import BSpline
        from .implicit import FieldListImplicitExtraction
        env = os.environ.get("TIGAR_IMPLICIT_M")
        if env == "0" or (env != "1" and self.comm.size == 1):
            return None
        kxs = []
        g0 = self.V.grids[0]
        for field in range(self.getNFields()):
            basis, grid = self.getScalarSpline(field), self.V.grids[field]
            if not (isinstance(basis, BSpline) and type(basis).getNodesAndEvals is BSpline.getNodesAndEvals) or \
                    not isinstance(grid, TensorNodeGrid) or grid.dg or grid.degree != g0.degree or \
                    any(not numpy.array_equal(a, b) for a, b in zip(grid.axes, g0.axes)):
                return None
            kx = self._kron_tables(basis, grid)
            if kx is None or not kx.products_stay_above(self.getIgnoreEps()):
                return None
            s_last = basis.splines[-1]
            kn, pl = numpy.asarray(s_last.knots, dtype=float), int(s_last.p)
            if not (numpy.all(kn[:pl + 1] == kn[0]) and numpy.all(kn[-(pl + 1):] == kn[-1])):
                return None
            kxs.append(kx)
        return FieldListImplicitExtraction(kxs, self.getIgnoreEps())

    def feRowOwners(self, nparts):
        """Rank of every FE row when the FE side is cut into ``nparts`` pieces the way this package cuts it: every
        field's nodes in ``nparts`` contiguous runs of (nearly) equal length -- whole node planes first for a
        tensor grid -- and piece r of every field on rank r.  (The reference takes the ownership from dolfin's mesh
        partition, tIGAr/common.py:1598-1600, 1612-1618.)"""
        owners = []
        for field in range(self.getNFields()):
            n = self.V.grids[field].num_nodes()
            owners.append((numpy.arange(n, dtype=numpy.int64) * int(nparts)) // max(n, 1))
        return numpy.concatenate(owners).astype(numpy.int32)

    def generatePermutation(self, nparts=None, fe_owner=None):
        """Order of the IGA dofs that follows the partition of the FE rows (tIGAr/common.py:1583-1665): every dof
        goes to the rank that owns most of the FE nodes of its support (``scipy.stats.mode`` there: the lowest
        rank on a tie; the support is everything ``getNodesAndEvals`` returns, zeros included), and the dofs are
        sorted by that rank.  Returns ``perm`` with new dof j = old dof perm[j] (the argsort the reference hands to
        MatPermute); the sort is stable, so within a rank the dofs keep their order.

        ``nparts``: number of ranks the extraction is prepared for (default: the size of the communicator);
        ``fe_owner``: owner of every FE row when it is not ``feRowOwners(nparts)``.  One part, or an operator that
        is never stored (tensor patch in z-slabs: the slabs of dofs already follow the slabs of FE rows): identity."""
        total = sum(self.getNcp(i) for i in range(self.getNFields()))
        nparts = int(self.comm.size if nparts is None else nparts)
        if fe_owner is not None:
            fe_owner = numpy.asarray(fe_owner, dtype=numpy.int32)
            nparts = max(nparts, int(fe_owner.max()) + 1 if fe_owner.size else 1)
        if nparts <= 1 or getattr(self.M, "is_implicit", False):
            return generateIdentityPermutation((0, total), self.comm)
        if fe_owner is None:
            fe_owner = self.feRowOwners(nparts)
        owner = self.generateM(support_only=True).transpose().majority_owner(fe_owner, nparts)
        return numpy.argsort(owner, kind="stable").astype(INDEX_TYPE)


class AbstractScalarBasis(object):
    """Scalar basis functions on a manifold with unique coordinates
    (tIGAr/common.py:1673-1759)."""

    __metaclass__ = abc.ABCMeta

    @abc.abstractmethod
    def getNodesAndEvals(self, xi):
        return

    @abc.abstractmethod
    def getNcp(self):
        return

    @abc.abstractmethod
    def generateMesh(self, comm=worldcomm):
        return

    @abc.abstractmethod
    def getDegree(self):
        return

    def needsDG(self):
        return True

    @abc.abstractmethod
    def useRectangularElements(self):
        return

    def getPrealloc(self):
        return DEFAULT_PREALLOC


class AbstractControlMesh(object):
    """Mapping from parametric to physical space (tIGAr/common.py:1762-1791)."""

    __metaclass__ = abc.ABCMeta

    @abc.abstractmethod
    def getHomogeneousCoordinate(self, node, direction):
        return

    @abc.abstractmethod
    def getScalarSpline(self):
        return

    @abc.abstractmethod
    def getNsd(self):
        return


class AbstractMultiFieldSpline(AbstractCoordinateChartSpline):
    """General multi-field spline built from AbstractScalarBasis members
    (tIGAr/common.py:1794-1885)."""

    __metaclass__ = abc.ABCMeta

    @abc.abstractmethod
    def getControlMesh(self):
        return

    @abc.abstractmethod
    def getFieldSpline(self, field):
        return

    def getPrealloc(self, control):
        if control:
            return self.getScalarSpline(-1).getPrealloc()
        return max(self.getScalarSpline(i).getPrealloc() for i in range(self.getNFields()))

    def getScalarSpline(self, field):
        if field == -1:
            return self.getControlMesh().getScalarSpline()
        return self.getFieldSpline(field)

    def getNsd(self):
        return self.getControlMesh().getNsd()

    def getHomogeneousCoordinate(self, node, direction):
        return self.getControlMesh().getHomogeneousCoordinate(node, direction)

    def getNodesAndEvals(self, x, field):
        return self.getScalarSpline(field).getNodesAndEvals(x)

    def generateMesh(self):
        return self.getScalarSpline(-1).generateMesh(comm=self.comm)

    def getDegree(self, field):
        return self.getScalarSpline(field).getDegree()

    def getNcp(self, field):
        return self.getScalarSpline(field).getNcp()

    def useDG(self):
        for i in range(-1, self.getNFields()):
            if self.getScalarSpline(i).needsDG():
                return True
        return False

    def _fieldGrid(self, field, dg):
        basis = self.getScalarSpline(field)
        deg = self.getDegree(field)
        try:
            return basis.generateMesh(comm=self.comm, degree=deg, dg=dg)
        except TypeError:
            grid = basis.generateMesh(comm=self.comm)
            if not isinstance(grid, TensorNodeGrid):
                raise TypeError("scalar bases used with tigar_amd must return a TensorNodeGrid "
                                "from generateMesh()")
            return grid


class EqualOrderSpline(AbstractMultiFieldSpline):
    """All unknown fields discretised with the control mesh's scalar basis
    (tIGAr/common.py:1891-1946).  ``EqualOrderSpline(numFields, controlMesh)``."""

    def customSetup(self, args):
        self.numFields = args[0]
        self.controlMesh = args[1]

    def getNFields(self):
        return self.numFields

    def getControlMesh(self):
        return self.controlMesh

    def getFieldSpline(self, field):
        return self.getScalarSpline(-1)

    def _single_shared_field(self):
        return self.numFields == 1

    def _fields_on_control_basis(self):
        return True

    def generateM(self, support_only=False):
        # one unknown field on the control mesh's basis: M is M_control (same rows, same
        # columns) -- share the device object instead of building it twice
        if self.numFields == 1 and getattr(self, "M_control", None) is not None and not support_only:
            return self.M_control
        return AbstractMultiFieldSpline.generateM(self, support_only)

    def addZeroDofsByLocation(self, subdomain, field):
        """Homogeneous Dirichlet BCs on the DoFs of ``field`` whose control points lie in
        ``subdomain`` (object with ``inside(x, on_boundary)``); tIGAr/common.py:1915-1946."""
        P = self._homogeneousCoordinateArray()
        nsd = self.getNsd()
        for I in range(P.shape[0]):
            p = P[I, 0:nsd] / P[I, nsd]
            if subdomain.inside(p, False) or subdomain.inside(p, True):
                self.zeroDofs += [self.globalDof(field, I)]


class FieldListSpline(AbstractMultiFieldSpline):
    """Multi-field spline from a list of scalar bases (tIGAr/common.py:1949-1970).
    ``FieldListSpline(controlMesh, fields)``."""

    def customSetup(self, args):
        self.controlMesh = args[0]
        self.fields = args[1]

    def getNFields(self):
        return len(self.fields)

    def getControlMesh(self):
        return self.controlMesh

    def getFieldSpline(self, field):
        return self.fields[field]


# ---- solver seam (b-4) ---------------------------------------------------------------------
class PETScKrylovSolver(object):
    """Look-alike of dolfin's ``PETScKrylovSolver(method, preconditioner)`` running on the
    GPU; plugs into ``ExtractedSpline.linearSolver`` (tIGAr/common.py:1255-1258), as in
    ``PETScKrylovSolver("gmres","jacobi")`` at demos/taylor-green/taylor-green-3d.py:89-90.
    Defaults are dolfin's [ext]: rtol 1e-6, atol 1e-15, maxit 10000, error on
    non-convergence."""

    # dolfin's names [ext: krylov_solver_methods() / krylov_solver_preconditioners()] -> what runs here.  Methods: cg,
    # gmres, bicgstab as named; the others that dolfin lists map onto the closest of the three.  Preconditioners: "none",
    # "jacobi" as named; "chebyshev" = a fixed Chebyshev polynomial in D^-1 K (csrc/tg_krylov.hip, tg_pcg_cheb), which also
    # stands in for the names whose PETSc implementation is a triangular sweep or a factorisation (sor, ilu, icc, bjacobi
    # = ILU(0) per process, amg ...): the request is honoured with the strongest preconditioner this library has for the
    # method, and ``preconditioner_requested`` / ``note`` say so -- dolfin users pass those names through the reference's
    # seam (tIGAr/common.py:1255-1258, 1292-1302) and should get a solve, not a ValueError.
    METHODS = {"cg": "cg", "gmres": "gmres", "bicgstab": "bicgstab", "default": "gmres", "minres": "gmres", "tfqmr": "bicgstab",
               "richardson": "gmres"}
    STRONG_PCS = ("sor", "ilu", "icc", "bjacobi", "amg", "hypre_amg", "petsc_amg", "hypre_euclid", "hypre_parasails", "ml_amg")

    def __init__(self, method="cg", preconditioner="jacobi", comm=None):
        self.method_requested, self.preconditioner_requested = method, preconditioner
        if method not in self.METHODS:
            raise ValueError("unknown Krylov method %r (%s)" % (method, ", ".join(sorted(self.METHODS))))
        method = self.METHODS[method]
        note = None
        if preconditioner == "default":
            preconditioner = "jacobi"
        if preconditioner in self.STRONG_PCS:
            note = " (preconditioner %r requested: " % (preconditioner,)
            if method == "cg":
                preconditioner = "chebyshev"
                note += "the Chebyshev polynomial preconditioner stands in for it)"
            else:
                preconditioner = "jacobi"
                note += "Jacobi stands in for it with %s)" % method
        if preconditioner not in ("none", "jacobi", "chebyshev"):
            raise ValueError("unknown preconditioner %r (none, jacobi, chebyshev, or one of dolfin's: %s)"
                             % (preconditioner, ", ".join(self.STRONG_PCS)))
        if preconditioner == "chebyshev" and method != "cg":
            raise ValueError("the Chebyshev polynomial preconditioner serves cg (a fixed symmetric polynomial)")
        self.method, self.preconditioner = method, preconditioner
        if method != self.method_requested and self.method_requested != "default":
            note = (note or "") + " (method %r requested: %s runs in its place)" % (self.method_requested, method)
        if note:
            # the request is honoured with a stand-in, and says so where the user sees it, not only in the
            # message of a failed solve
            import warnings
            warnings.warn("PETScKrylovSolver(%r, %r): running (%r, %r)%s" % (self.method_requested, self.preconditioner_requested,
                                                                            method, preconditioner, note), stacklevel=2)
        self.parameters = {"relative_tolerance": 1e-6, "absolute_tolerance": 1e-15,
                           "maximum_iterations": 10000, "error_on_nonconvergence": True,
                           "nonzero_initial_guess": False, "gmres_restart": 30,
                           # (not a dolfin parameter) products per application of the Chebyshev preconditioner
                           "chebyshev_degree": 8,
                           "report": False, "monitor_convergence": False,
                           # (not a dolfin parameter) give up with status -3 after 25 GMRES restart cycles without
                           # progress instead of running to the iteration limit as PETSc does; set by the default
                           # solver that stands in for the reference's direct LU on large systems
                           "stagnation_guard": False}
        self.comm = comm
        self.last = None
        self.note = note       # appended to the non-convergence message (who chose this solver / what stands in)

    REASONS = {-1: "iteration limit reached", -2: "breakdown (NaN / zero pivot in the recurrence)",
               -3: "stagnation (25 GMRES restart cycles without progress)"}

    def solve(self, A, x, b):
        """``x`` is the output; with ``parameters["nonzero_initial_guess"]`` it also holds the start vector
        (the convergence test stays relative to ||B b||, PETSc's default [ext])."""
        A, x, b = _as_device_csr(A), _as_device_vector(x), _as_device_vector(b)
        guess = bool(self.parameters["nonzero_initial_guess"])

        def run(pc):
            return _dev.krylov_solve(
                A, b, x, self.method, pc, self.parameters["relative_tolerance"],
                self.parameters["absolute_tolerance"], self.parameters["maximum_iterations"],
                self.parameters["chebyshev_degree"] if pc == "chebyshev" else self.parameters["gmres_restart"],
                self.comm, nonzero_initial_guess=guess,
                stagnation_guard=bool(self.parameters.get("stagnation_guard", False)),
                # (a matrix this package assembled from a form that is symmetric by construction: the half-storage copy
                #  of a CG solve is not compared with the CSR product again, csrc/tg_symgrid.hip)
                symmetric=bool(getattr(A, "symmetric_by_construction", False)))

        x0 = None
        if self.preconditioner == "chebyshev" and guess:
            x0 = DeviceVector(x.size())
            _dev.vec_copy_range(x0, 0, x, 0, x.size())
        its, res, status = run(self.preconditioner)
        fallback = None
        if status == -2 and self.preconditioner == "chebyshev":
            # the polynomial is built on an ESTIMATE of the largest eigenvalue of D^-1 K (a few power iterations); an estimate
            # that is too small makes the polynomial indefinite and the recurrence breaks down although K is fine.
            # Jacobi-CG needs no estimate: run it instead of reporting a breakdown.  A K or b that holds NaN breaks
            # it down as well and the status stays -2.
            import warnings
            warnings.warn("PETScKrylovSolver: Chebyshev-preconditioned CG broke down (eigenvalue estimate); "
                          "solving with Jacobi-CG instead", stacklevel=2)
            if x0 is not None:
                _dev.vec_copy_range(x, 0, x0, 0, x.size())
            its_c = its
            its, res, status = run("jacobi")
            fallback = {"preconditioner": "jacobi", "after_iterations": its_c}
        self.last = {"iterations": its, "residual_norm": res, "status": status, "method": self.method,
                     "preconditioner": fallback["preconditioner"] if fallback else self.preconditioner,
                     "method_requested": self.method_requested, "preconditioner_requested": self.preconditioner_requested}
        if fallback:
            self.last["fallback"] = fallback
        if status < 0 and self.parameters["error_on_nonconvergence"]:
            raise RuntimeError("Krylov solver (%s, %s) did not converge: %s after %d iterations, preconditioned "
                               "residual %.3e.%s" % (self.method, self.preconditioner,
                                                     self.REASONS.get(status, "status %d" % status), its, res,
                                                     self.note or ""))
        return its


KrylovSolver = PETScKrylovSolver


class PETScLUSolver(object):
    """Look-alike of dolfin's ``PETScLUSolver`` / ``LUSolver``: direct solve on the GPU by banded LU with partial
    pivoting (``tg_lu_solve``, LAPACK dgbtrf's storage and pivoting).  Plugs into ``ExtractedSpline.linearSolver``.
    IGA matrices of tensor-product patches are banded in the patch numbering; for other orderings (several fields
    numbered field after field) ``parameters["reorder"]`` (default "auto") applies a reverse Cuthill-McKee
    permutation of the pattern first.  Refuses systems whose band storage exceeds ``max_band_bytes``."""

    def __init__(self, method="default", comm=None):
        self.parameters = {"reorder": "auto", "max_band_bytes": 16 * 2 ** 30, "report": False,
                           "symmetric": False, "reuse_factorization": False}
        self.last = None

    def band_cost(self, A):
        """(bytes of band storage, multiply-adds) of factorising ``A`` as it is numbered"""
        kl, ku, nb = _dev.lu_band_info(A)
        return nb, 2.0 * A.shape[0] * kl * (kl + ku)

    def solve(self, A, x, b):
        A, x, b = _as_device_csr(A), _as_device_vector(x), _as_device_vector(b)
        n = A.shape[0]
        kl, ku, nb = _dev.lu_band_info(A)
        perm = None
        mode = self.parameters["reorder"]
        if mode is True or (mode == "auto" and nb > 2 ** 28 and (kl + ku) > n // 8):
            # bandwidth-reducing ordering of the PATTERN on the host (symbolic step; values stay on the device
            # except for this one re-upload of the permuted matrix)
            import scipy.sparse as _sp
            from scipy.sparse.csgraph import reverse_cuthill_mckee
            S = A.to_scipy()
            # (the STORED pattern: entries zeroed by MatZeroRowsColumns stay in the band storage, so they must stay in
            #  the graph -- scipy's sum of abs(S) would drop them and leave the boundary dofs isolated)
            ones = _sp.csr_matrix((numpy.ones(S.nnz, dtype=numpy.int8), S.indices, S.indptr), shape=S.shape)
            pat = (ones + ones.T).tocsr()
            prm = numpy.asarray(reverse_cuthill_mckee(pat, symmetric_mode=True), dtype=numpy.int64)
            Sp = S[prm][:, prm].tocsr()
            Ap = DeviceCSR.from_scipy(Sp)
            kl2, ku2, nb2 = _dev.lu_band_info(Ap)
            if nb2 < nb:
                A, perm, kl, ku, nb = Ap, prm, kl2, ku2, nb2
        if nb > self.parameters["max_band_bytes"]:
            raise MemoryError("direct solve: the band storage of this %d x %d system (half-bandwidths %d / %d) needs %.1f GB; "
                              "use a Krylov solver (PETScKrylovSolver) for systems of this size" % (n, n, kl, ku, nb / 2 ** 30))
        chol0 = _dev.prof_get(8)[1]
        if perm is not None:
            bp = DeviceVector(data=b.get_local()[perm])
            xp = DeviceVector(n)
            info = _dev.lu_solve(A, bp, xp)
            if info == 0:
                out = numpy.empty(n)
                out[perm] = xp.get_local()
                x.set_local(out)
        else:
            info = _dev.lu_solve(A, b, x)
        # (a symmetric positive definite system is factorised as L L^T -- blocked banded Cholesky on the matrix cores,
        #  csrc/tg_chol.hip -- instead of P A = L U; TIGAR_LU_CHOLESKY=0: always the LU)
        self.last = {"info": info, "kl": kl, "ku": ku, "band_bytes": nb, "reordered": perm is not None,
                     "factorisation": "cholesky" if _dev.prof_get(8)[1] > chol0 else "lu"}
        if info != 0:
            raise RuntimeError("direct solve: the matrix is singular (exact zero pivot in column %d)" % (info - 1))
        return 1


LUSolver = PETScLUSolver


class _DefaultSolver(object):
    """What runs when ``linearSolver`` is None.  The reference calls dolfin's ``solve`` there, i.e. a sparse direct LU
    (tIGAr/common.py:1255-1256 [ext]) -- every demo relies on it.  Here: the banded direct solver above whenever its
    cost is moderate (band storage <= 8 GB and <= 4e12 multiply-adds: all 2-D patches of the demos, small 3-D ones),
    otherwise Jacobi-preconditioned GMRES(30) (CG for the normal equations of ``FEtoIGA``) to a relative residual of
    1e-12, bounded by PETSc's default 10 000 iterations and by the solver's stagnation guard, with an error message
    that names the deviation when it gives up."""

    def __init__(self, method="gmres"):
        self.method = method
        self.comm = None
        self.last = None

    def solve(self, A, x, b):
        A = _as_device_csr(A)
        if self.comm is None and os.environ.get("TIGAR_DEFAULT_SOLVER", "auto") != "krylov":
            lu = PETScLUSolver()
            kl, ku, nb = _dev.lu_band_info(A)
            flops = 2.0 * A.shape[0] * kl * (kl + ku)
            fits = nb <= 8 * 2 ** 30 and flops <= 4e12 and A.shape[0] <= 400000
            # Beyond the LU's limits (3-D patches): a symmetric positive definite system -- the common case -- still gets
            # a direct solve, by the banded Cholesky factorisation: a quarter of the LU's multiply-adds, on the matrix
            # cores, a third of its band (csrc/tg_chol.hip; it declines what is not SPD and the next choice runs).
            chol_fits = kl == ku and 8.0 * A.shape[0] * (kl + 1) <= 96 * 2 ** 30 and float(A.shape[0]) * kl * kl <= 4e13 \
                and kl <= 16000 and os.environ.get("TIGAR_LU_CHOLESKY", "1") != "0"

            def cholesky():
                xd, bd = _as_device_vector(x), _as_device_vector(b)
                if not _dev.chol_solve(A, bd, xd):
                    return False
                self.last = {"solver": "lu", "factorisation": "cholesky", "info": 0, "kl": kl, "ku": ku,
                             "band_bytes": 8 * A.shape[0] * (kl + 1), "reordered": False}
                return True

            # a band as numbered (kl well below n: one field) goes there at once; a field-major system of several fields
            # (kl ~ n (nF-1)/nF) has its reordered band evaluated first (a download of K and a host ordering: 0.2-0.45 s for
            # 15-34 M entries, which the single-field solves paid for nothing)
            chol_tried = False
            if not fits and chol_fits and 4 * kl < A.shape[0]:
                if cholesky():
                    return 1
                chol_tried = True
            if not fits and A.shape[0] <= 400000 and A.nnz <= 2e8:
                # as numbered the band is too wide -- field-major systems of several fields (kl ~ n (nF-1)/nF): the saddle
                # point and elasticity cases where the reference's direct solver matters.  Evaluate the band of the
                # reverse Cuthill-McKee ordering of the pattern before giving LU up.
                import scipy.sparse as _sp
                from scipy.sparse.csgraph import reverse_cuthill_mckee
                S = A.to_scipy()
                pat = _sp.csr_matrix((numpy.ones(S.nnz, dtype=numpy.int8), S.indices, S.indptr), shape=S.shape)
                prm = numpy.asarray(reverse_cuthill_mckee((pat + pat.T).tocsr(), symmetric_mode=True), dtype=numpy.int64)
                inv = numpy.empty_like(prm)
                inv[prm] = numpy.arange(prm.size)
                coo = S.tocoo()
                dist = inv[coo.row] - inv[coo.col]
                kl2, ku2 = int(max(0, dist.max())), int(max(0, -dist.min()))
                nb2 = 8.0 * A.shape[0] * (2 * kl2 + ku2 + 1)
                if nb2 <= 8 * 2 ** 30 and 2.0 * A.shape[0] * kl2 * (kl2 + ku2) <= 4e12:
                    fits = True
                    lu.parameters["reorder"] = True
            if fits or os.environ.get("TIGAR_DEFAULT_SOLVER") == "lu":
                lu.solve(A, x, b)
                self.last = dict(lu.last, solver="lu")
                return 1
            if chol_fits and not chol_tried and cholesky():
                return 1
        ks = PETScKrylovSolver(self.method, "jacobi", comm=self.comm)
        ks.parameters["relative_tolerance"] = 1e-12
        ks.parameters["maximum_iterations"] = 10000
        ks.parameters["stagnation_guard"] = True
        ks.note = ("  (linearSolver=None: the reference would have run dolfin's direct LU here; this system is too large for "
                   "tigar_amd's banded direct solver, so Jacobi-%s was used -- set ExtractedSpline.linearSolver for "
                   "ill-conditioned systems)" % self.method.upper())
        ks.solve(A, x, b)
        self.last = dict(ks.last, solver=self.method)
        return ks.last["iterations"]


def _default_linear_solver(method="gmres"):
    return _DefaultSolver(method)


# ---- analysis side ---------------------------------------------------------------------------
class ExtractedSpline(object):
    """
    An extracted spline (tIGAr/common.py:667-1433), restricted to the hot path: the
    extraction operators, their application to FE matrices/vectors, and the linear solve.
    UFL form construction (``grad``, ``dx``, ``rationalize`` ...) needs FEniCS and is not
    provided here.
    """

    def __init__(self, sourceArg, quadDeg=None, mesh=None, doPermutation=DEFAULT_DO_PERMUTATION,
                 comm=worldcomm):
        if isinstance(sourceArg, AbstractExtractionGenerator):
            self.initFromGenerator(sourceArg, quadDeg, doPermutation)
        else:
            self.initFromFilesystem(sourceArg, quadDeg, comm, mesh)
        self.genericSetup()

    def initFromFilesystem(self, dirname, quadDeg, comm, mesh=None):
        """Instance from the extraction data in directory ``dirname`` (tIGAr/common.py:748-894):
        ``extraction-info.txt``, the PETSc binary matrices and index set, and ``extraction-data.h5`` (control functions
        from ``/control<i>/vector_0``, node sets from the group ``/tigar_amd``; see ``writeExtraction``).  M is then an
        arbitrary sparse matrix: extractMatrix uses the general PtAP kernel."""
        from . import petscio
        self.quadDeg = quadDeg
        self.comm = comm
        with open(os.path.join(dirname, EXTRACTION_INFO_FILE), "r") as f:
            lines = f.read().split("\n")
        self.nsd = int(lines[0])
        self.elementType = lines[1]
        self.nFields = int(lines[2])
        self.p_control = int(lines[3])
        ncp_control = int(lines[4])
        self.p, ncp = [], []
        for i in range(self.nFields):
            self.p.append(int(lines[5 + 2 * i]))
            ncp.append(int(lines[6 + 2 * i]))
        h5_path, npz_path = os.path.join(dirname, EXTRACTION_DATA_FILE), os.path.join(dirname, EXTRACTION_DATA_NPZ)
        h5 = None
        if os.path.exists(h5_path):
            from . import h5io
            h5 = h5io.H5File(h5_path, "r")
            if not h5.exists(EXTRACTION_H5_PRIVATE):
                h5.close()
                raise IOError("%s carries no group %s: it was not written by this package.  Directories written by the "
                              "reference hold M in dolfin's own dof numbering, which cannot be matched to node positions "
                              "without dolfin" % (h5_path, EXTRACTION_H5_PRIVATE))
            data = _H5Archive(h5)
            control = [h5.read_dataset(EXTRACTION_H5_CONTROL_FUNC_NAME(i) + "/vector_0") for i in range(self.nsd + 1)]
        elif os.path.exists(npz_path):                     # directories written by rounds 1-2 of this package
            data = numpy.load(npz_path)
            control = [data["control%d" % i] for i in range(self.nsd + 1)]
        else:
            raise IOError("neither %s nor %s found" % (h5_path, npz_path))

        def space(name):
            grids = []
            for gi in range(int(data[name + "_ngrids"])):
                grids.append(_grid_from_arrays(data, "%s_%d" % (name, gi)))
            return TensorFunctionSpace(grids, str(data[name + "_element"]))
        self.mesh = mesh
        self.V_control = space("control")
        self.V = space("fields")
        self.cpFuncs = []
        for i in range(self.nsd + 1):
            f = Function(self.V_control)
            f.vector().set_local(control[i])
            self.cpFuncs.append(f)
        if h5 is not None:
            h5.close()
        self.M_control = DeviceCSR.from_scipy(petscio.read_mat(os.path.join(dirname, EXTRACTION_MAT_FILE_CTRL)))
        self.M = DeviceCSR.from_scipy(petscio.read_mat(os.path.join(dirname, EXTRACTION_MAT_FILE)))
        if self.M_control.shape != (self.V_control.dim(), ncp_control) or self.M.shape != (self.V.dim(), sum(ncp)):
            raise ValueError("extraction matrices in %s do not match extraction-info.txt" % dirname)
        self.zeroDofs = numpy.asarray(petscio.read_is(os.path.join(dirname, EXTRACTION_ZERO_DOFS_FILE)), dtype=INDEX_TYPE)
        self._kron = None

    def initFromGenerator(self, generator, quadDeg, doPermutation=DEFAULT_DO_PERMUTATION):
        """tIGAr/common.py:708-746 -- shares M, M_control, V, cpFuncs with the generator."""
        if doPermutation:
            generator.applyPermutation()
        self.quadDeg = quadDeg
        self.nsd = generator.getNsd()
        self.elementType = generator.extractionElement()
        self.nFields = generator.getNFields()
        self.p_control = generator.getDegree(-1)
        self.p = [generator.getDegree(i) for i in range(self.nFields)]
        self.mesh = generator.mesh
        self.cpFuncs = generator.cpFuncs
        self.V = generator.V
        self.V_control = generator.V_control
        self.M = generator.M
        self.M_control = generator.M_control
        self.comm = generator.getComm()
        self._generator_engine = getattr(generator, "_slab_engine", None)
        self._generator = generator
        self._kron = getattr(generator, "_kron", None)
        self._kron_scalar = getattr(generator, "_kron_scalar", None)
        self._kron_fields = getattr(generator, "_kron_fields", None)
        self.zeroDofs = generator.zeroDofsArray().astype(INDEX_TYPE)

    def genericSetup(self):
        self.setSolverOptions()
        self.MT = self.M.transpose()          # explicit M^T (cached on M by the generator)
        self._ptap_plan = None
        self._ptap_plan_key = None
        self._slab = None

    # -- streamed / distributed engine -------------------------------------------------------------
    def _implicit(self):
        return bool(getattr(self.M, "is_implicit", False))

    def _distributed(self):
        return self.comm is not None and getattr(self.comm, "size", 1) > 1

    def _slab_path(self):
        """The z-slab engine (``tigar_amd.dist.SlabHotPath``) behind extractMatrix / extractVector /
        solveLinearSystem when M is implicit or the patch is spread over several ranks: rows of K in
        sub-slabs of dof planes, only K resident (PETSc's row-block MatPtAP, tIGAr/common.py:1194-1195)."""
        if self._slab is None and getattr(self, "_generator_engine", None) is not None and self.nFields == 1:
            self._slab = self._generator_engine
        if self._slab is None and self._kron is None and getattr(self, "_kron_scalar", None) is not None:
            # several fields on one tensor basis: the scalar engine per block, fields interleaved plane by plane
            from .dist import FieldSlabPath
            kx = self._kron_scalar
            dc = self.comm.device() if self._distributed() else None
            self._slab = FieldSlabPath(kx.basis, kx.grid, self.nFields, self.comm.rank if dc is not None else 0,
                                       self.comm.size if dc is not None else 1, dc, sub_planes="auto",
                                       eps=getattr(self.M, "eps", DEFAULT_BASIS_FUNC_IGNORE_EPS), kx=kx)
        if self._slab is None and self._kron is None and getattr(self, "_kron_fields", None) is not None:
            # fields on different tensor bases over one node grid: one split of the plane index, pair walks per block
            from .dist import FieldListSlabPath
            dc = self.comm.device() if self._distributed() else None
            self._slab = FieldListSlabPath(self._kron_fields, self.comm.rank if dc is not None else 0,
                                           self.comm.size if dc is not None else 1, dc, sub_planes="auto",
                                           eps=getattr(self.M, "eps", DEFAULT_BASIS_FUNC_IGNORE_EPS))
        if self._slab is None:
            if self._kron is None:
                raise NotImplementedError("the streamed / multi-GPU path needs tensor-product B-spline fields (one basis, or "
                                          "several bases over one node grid)")
            from .dist import SlabHotPath
            kx = self._kron
            dc = self.comm.device() if self._distributed() else None
            self._slab = SlabHotPath(kx.basis, kx.grid, self.comm.rank if dc is not None else 0,
                                     self.comm.size if dc is not None else 1, dc, sub_planes="auto",
                                     eps=getattr(self.M, "eps", DEFAULT_BASIS_FUNC_IGNORE_EPS), kx=kx)
        return self._slab

    def localDofRange(self):
        """IGA dofs [g0, g1) owned by this rank (PETSc getOwnershipRange of MTAM's rows)."""
        if self._distributed() or (self._implicit() and self.nFields > 1):
            return self._slab_path().mine["dofs"]
        return (0, self.M.shape[1])

    def localDofIndices(self):
        """Reference (field-after-field) index of every entry of this rank's IGA vectors and rows of MTAM, in local order.
        One field: the contiguous range ``localDofRange()``.  Several fields on several ranks: the dofs are renumbered
        plane by plane across the fields so that a rank's block stays contiguous (``tigar_amd.dist.FieldSlabPath``) -- as
        the reference renumbers the IGA dofs for parallel runs (generatePermutation, tIGAr/common.py:1583-1665)."""
        sp_ = self._slab_path() if (self._distributed() or self._implicit()) else None
        if sp_ is not None and hasattr(sp_, "local_dof_indices"):
            return sp_.local_dof_indices()
        g0, g1 = self.localDofRange()
        return numpy.arange(g0, g1, dtype=numpy.int64)

    def localFERange(self):
        """FE rows whose prolongation u = M U this rank computes (several fields: one range per field)."""
        if self._distributed() or (self._implicit() and self.nFields > 1):
            return self._slab_path().mine["u_rows"]
        return (0, self.M.shape[0])

    # -- a-10
    def extractVector(self, b, applyBCs=True):
        """Apply extraction to an FE vector ``b``: ``M^T b``, zeroed at ``zeroDofs`` if
        ``applyBCs`` (tIGAr/common.py:1142-1160)."""
        from .implicit import LazyFEVector
        if isinstance(b, LazyFEVector) or self._distributed() or (self._implicit() and self.nFields > 1):
            # (several fields with an implicit operator: the field-block engine and its plane-wise numbering, also on
            #  one rank -- K, M^T b and U must share it)
            if isinstance(b, LazyFEVector):
                rows = b.rows
            else:
                full = _as_device_vector(b)       # replicated FE vector: every rank takes its rows

                def rows(r0, r1):
                    piece = DeviceVector(r1 - r0)
                    _dev.vec_copy_range(piece, 0, full, r0, r1 - r0)
                    return piece
            return self._slab_path().assemble_vector(rows, self.zeroDofs if applyBCs else None,
                                                     getattr(self, "stage_timers", None))
        MTb = self.M.mult_transpose(_as_device_vector(b))
        if applyBCs:
            MTb.zero_entries(self.zeroDofs)
        return MTb

    def assembleVector(self, form, applyBCs=True):
        """``form``: anything with ``.assemble_vector(V)`` (see ``tigar_amd.forms``) or an
        already assembled FE vector (tIGAr/common.py:1162-1173).  With several ranks the form is
        asked for the FE rows each rank needs (``assemble_vector(V, row0, row1)``)."""
        if not self.__dict__.get("_in_system"):
            self._new_ghost_epoch()
        if hasattr(form, "assemble_vector"):
            if self._distributed():
                from .implicit import LazyFEVector
                b = LazyFEVector(lambda r0, r1: form.assemble_vector(self.V, r0, r1), self.V.dim())
            else:
                b = form.assemble_vector(self.V)
        else:
            b = form
        return self.extractVector(b, applyBCs=applyBCs)

    # -- a-11
    def extractMatrix(self, A, applyBCs=True, diag=1):
        """Apply extraction to an FE matrix ``A``: ``M^T A M`` (PtAP), then rows and columns
        of ``zeroDofs`` zeroed with ``diag`` on the diagonal (tIGAr/common.py:1176-1204).
        The symbolic plan is cached and reused while A's pattern is unchanged."""
        from .implicit import LazyFEMatrix
        zd = self.zeroDofs if applyBCs else None
        if isinstance(A, LazyFEMatrix) or self._distributed() or (self._implicit() and self.nFields > 1):
            # (several fields with an implicit operator, also on one rank: the field-block engine and its plane-wise
            #  numbering -- the same guard as extractVector / solveLinearSystem, so that K, M^T b and U share it;
            #  an explicit A -- FEtoIGA's identity, an uploaded matrix -- is cut into the blocks the engine asks for)
            a_fac = None
            if self.nFields > 1 and self._kron is None and (getattr(self, "_kron_scalar", None) is not None or
                                                            getattr(self, "_kron_fields", None) is not None):
                return self._slab_path().assemble_matrix(self._block_producer(A), zd, float(diag),
                                                         getattr(self, "stage_timers", None),
                                                         block_factors=getattr(A, "block_factors", None))
            if isinstance(A, LazyFEMatrix):
                a_rows = A.rows
                a_fac = A.kron_factors
            else:
                # an assembled FE matrix handed to every rank (the reference's A is a distributed PETSc matrix whose
                # rows MatPtAP redistributes, tIGAr/common.py:1194-1195): every rank cuts the row blocks of its slab out
                # of its copy -- on the device when it is a DeviceCSR, on the host (then uploaded) when it is scipy
                a_rows = self._row_blocks_of(A)
            return self._slab_path().assemble_matrix(a_rows, zd, float(diag), getattr(self, "stage_timers", None),
                                                     a_factors=a_fac)
        A = _as_device_csr(A)
        by_blocks = self._kron is None and getattr(self, "_kron_scalar", None) is not None
        # 2-D tensor patches (one or several fields on one basis): the whole product in two line-walk passes when A
        # carries the element-coupling pattern (verified on the device; csrc/tg_tensor_body.h)
        kx2 = self._kron if self._kron is not None else getattr(self, "_kron_scalar", None)
        if kx2 is not None and kx2.d == 2 and not A.is_loose() and os.environ.get("TIGAR_PTAP_FACTORED", "1") != "0":
            from .tensorptap import TensorPtAP2D
            nF2 = self.nFields if by_blocks else 1
            plan2 = TensorPtAP2D.for_extraction(kx2, nF2)
            if plan2 is not None:
                K = plan2.ptap(A, zd, float(diag))
                if K is not None:
                    return K
            elif os.environ.get("TIGAR_PTAP_UNWRAP", "1") != "0":
                # periodic directions (tIGAr/BSplines.py:204-212): the walks on the space before the wrapped functions
                # are identified, then K = R^T K_u R (kronptap.KronExtraction.unwrapped / fold)
                ku2 = kx2.unwrapped()
                plan2 = TensorPtAP2D.for_extraction(ku2, nF2) if ku2 is not None else None
                K_u = plan2.ptap(A, None, 1.0) if plan2 is not None else None
                if K_u is not None:
                    return ku2.fold(K_u, zd, float(diag), nfields=nF2)
        if by_blocks and os.environ.get("TIGAR_PTAP_FACTORED", "1") != "0":
            K = self._extract_matrix_by_field_blocks(A, zd, float(diag))
            if K is not None:
                return K
        if self._kron is None and not by_blocks and getattr(self, "_kron_fields", None) is not None and \
                os.environ.get("TIGAR_PTAP_FACTORED", "1") != "0":
            K = self._extract_matrix_by_field_list(A, zd, float(diag))
            if K is not None:
                return K
        if self._kron is not None:
            from .kronptap import default_groups, ptap_factored
            kx = self._kron
            groups = default_groups(kx.d, max(s1.p for s1 in kx.basis.splines))
            if os.environ.get("TIGAR_PTAP_FACTORED", "1") != "0":
                # Kronecker-structured M: dense-box kernel (one stage or sum-factorised stages)
                stored = None if self._implicit() else (self.M, self.MT)
                return ptap_factored(kx, A, (0, kx.nfe[-1]), (0, kx.nfe[-1]), (0, kx.ncp[-1]), zd, float(diag), groups,
                                     stored=stored)
        if self._implicit():
            # an implicit operator that is not to be used as a Kronecker product (TIGAR_PTAP_FACTORED=0) with an assembled A: the
            # streamed engine materialises M chunk by chunk and takes its general stages -- element chunks, or the row-wise
            # kernels for a matrix they decline (round 6; until then: NotImplementedError)
            return self._slab_path().assemble_matrix(self._row_blocks_of(A), zd, float(diag), getattr(self, "stage_timers", None))
        # cell-local FE spaces (T-splines, multi-patch B-splines: meshes of disconnected cells): an assembled A is block
        # diagonal with one dense block per cell and the product is a sum of small dense triple products
        # (tigar_amd/cellptap.py); the plan depends on M only and is kept, A is verified on the device at every call
        if os.environ.get("TIGAR_PTAP_CELLS", "1") != "0":
            from .cellptap import CellBlockPtAP, block_size_of, cell_size_with_extras, split_cells
            b = block_size_of(A)
            extras = False
            if not b:
                b = cell_size_with_extras(A)
                extras = bool(b)
            if b and A.shape[0] == self.M.shape[0]:
                plans = self.__dict__.setdefault("_cell_plans", {})
                if b not in plans:
                    try:
                        plans[b] = CellBlockPtAP(self.M, b)
                    except ValueError:
                        plans[b] = None
                if plans[b] is not None and not extras:
                    K = plans[b].ptap(A, zd, float(diag))
                    if K is not None:
                        return K
                elif plans[b] is not None:
                    # couplings outside the cell blocks (contact / penalty terms added by hand: the reason extractMatrix takes
                    # any A, tIGAr/common.py:1175; demos/kl-shell-svk/reef-knot.py:455-467): A = D + R on the device, the
                    # dense blocks D through the cell-block product, the few entries of R through the general kernels, the
                    # two added on the union of their patterns (= the structural product of A), then MatZeroRowsColumns
                    parts = plans[b].ptap_extras(A)
                    if parts is not None:
                        KD, R = parts
                        if KD is not None:
                            from .cellptap import remainder_product
                            KR = remainder_product(R, self.M, self.__dict__.setdefault("_cellR_cache", {}))
                            self._cellR_key = ("cells-R", R.shape, R.nnz)
                            if KR is None:
                                KR = DeviceCSR.from_scipy(_scipy_zero(KD.shape))
                            K = KD.add(KR)
                            del KD, KR
                            if zd is not None and len(zd):
                                K.zero_rows_cols(numpy.asarray(zd, dtype=numpy.int32), float(diag))
                            return K
        # connected meshes (what dolfin assembles on the Q_p / P_p mesh of the extraction, with an M that is used as a general
        # CSR matrix): A split into one dense block per cell, then the same dense cell products (tigar_amd/elemptap.py)
        if os.environ.get("TIGAR_PTAP_ELEMENTS", "1") != "0" and not A.is_loose():
            K = self._extract_matrix_by_elements(A, zd, float(diag))
            if K is not None:
                return K
        key = (A.shape, A.nnz)
        fresh = self._ptap_plan is None or self._ptap_plan_key != key
        if fresh:
            self._ptap_plan = _dev.ptap_symbolic(A, self.M, self.MT)
            self._ptap_plan_key = key
        try:
            return _dev.ptap_numeric(self._ptap_plan, A, self.M, self.MT, zd, float(diag))
        except _dev.TigarHipError:
            if fresh and by_blocks:
                # rows of the whole product beyond the general kernels' per-row tables (three fields at p = 3 in 3-D):
                # the same kernels block by block, on the scalar operands
                self._ptap_plan = self._ptap_plan_key = None
                K = self._extract_matrix_by_field_blocks(A, zd, float(diag), tensor=False)
                if K is not None:
                    return K
            if fresh:
                raise
            # same shape and nnz but another sparsity pattern than the cached plan's: the reference
            # recomputes the symbolic product on every call (tIGAr/common.py:1194-1195) -- plan again
            self._ptap_plan = _dev.ptap_symbolic(A, self.M, self.MT)
            return _dev.ptap_numeric(self._ptap_plan, A, self.M, self.MT, zd, float(diag))

    def _extract_matrix_by_elements(self, A, zd, diag):
        """M^T A M by the element-split cell-block product, or None when it does not apply: one field on one mesh whose cells
        hold at most 125 nodes (the cells' node lists are the dofmap of ``self.V``), a system large enough for the plan to pay
        (``TIGAR_PTAP_ELEMENTS=2``: any size), every entry of A between nodes of a common cell (any assembled FE matrix;
        others fall through to the general kernels)"""
        grids = getattr(getattr(self, "V", None), "grids", None)
        if self.nFields != 1 or not grids or len(grids) != 1 or A.shape != (self.M.shape[0], self.M.shape[0]):
            return None
        plan = self.__dict__.get("_elem_plan")
        if plan is None or plan[0] is not self.M:
            plan = (self.M, self._element_plan_for(self.M))
            self.__dict__["_elem_plan"] = plan
        if plan[1] is None:
            return None
        return plan[1].ptap(A, zd, diag)

    def _element_plan_for(self, M):
        """ElementSplitPtAP for the scalar extraction operator ``M`` on the (first) mesh of the FE space, or None"""
        if os.environ.get("TIGAR_PTAP_ELEMENTS", "1") == "0":
            return None
        grids = getattr(getattr(self, "V", None), "grids", None)
        if not grids or (M.shape[0] < 20000 and os.environ.get("TIGAR_PTAP_ELEMENTS", "1") != "2"):
            return None
        g = grids[0]
        if (int(g.degree) + 1) ** g.dim() > 125 or int(g.degree) < 1 or getattr(g, "dg", False) or g.num_nodes() != M.shape[0]:
            return None
        from .elemptap import ElementSplitPtAP, CellNodes
        try:
            return ElementSplitPtAP(M, CellNodes.from_grid(g))       # (the dofmap of V, generated on the device)
        except (ValueError, _dev.TigarHipError):
            return None

    def _block_producer(self, A):
        """``a_block(f, g, r0, r1)`` for the field-block engine: rows [r0, r1) of block (f, g) of an FE matrix on the mixed
        space, columns of one field; None where the producer says the fields are not coupled"""
        from .implicit import LazyFEMatrix
        nfe = self.V.dim() // self.nFields
        if isinstance(A, LazyFEMatrix) and getattr(A, "block_rows", None) is not None:
            return A.block_rows
        rows = A.rows if isinstance(A, LazyFEMatrix) else self._row_blocks_of(A)

        def a_block(f, g, r0, r1):
            blk = rows(f * nfe + int(r0), f * nfe + int(r1))
            return blk.block(0, int(r1) - int(r0), g * nfe, (g + 1) * nfe)
        return a_block

    @staticmethod
    def _row_blocks_of(A):
        """``rows(r0, r1)`` of an explicit FE matrix: DeviceCSR row blocks with global columns"""
        if isinstance(A, DeviceCSR):
            ncols = A.shape[1]
            return lambda r0, r1: A.block(int(r0), int(r1), 0, ncols)
        import scipy.sparse as _sp
        if not _sp.issparse(A):
            raise TypeError("extractMatrix: a DeviceCSR, a scipy sparse matrix or a LazyFEMatrix is expected")
        Ah = _sp.csr_matrix(A)
        return lambda r0, r1: DeviceCSR.from_scipy(Ah[int(r0):int(r1)])

    def _extract_matrix_by_field_blocks(self, A, zd, diag, tensor=True):
        """M^T A M for several fields on one tensor basis (M = diag(M_s, ..., M_s), dofs field after field): block
        (i, j) of the result is M_s^T A_ij M_s, computed on the block cut out of A -- by the scalar tensor-pattern
        passes (csrc/tg_tensor_body.h) where the patch and the block qualify, by the general kernels on the scalar
        operands otherwise (whose per-row tables hold a scalar row's intermediate, not that of nFields of them).  The
        blocks are put together and MatZeroRowsColumns is applied to the whole (tIGAr/common.py:1194-1200).  None
        when A is not a matrix on this mixed space."""
        from .tensorptap import TensorPtAP
        kx = self._kron_scalar
        plan = TensorPtAP.for_extraction(kx) if tensor else None
        nF = self.nFields
        nfe = int(numpy.prod(kx.nfe, dtype=numpy.int64))
        ncp = int(numpy.prod(kx.ncp, dtype=numpy.int64))
        if A.shape != (nF * nfe, nF * nfe):
            return None
        nz, kz = int(kx.nfe[-1]), int(kx.ncp[-1])
        # FE planes per call of the x / y passes: their first intermediate is about 2.5 x the block's own bytes
        step = max(kx.basis.splines[-1].p, min(nz, int(2.0e10 // max(1.0, 12.0 * 2.5 * (A.nnz / float(nF * nF)) / nz))))
        scalar = {}

        def general(Aij):
            if tensor:
                # the scalar machinery for a Kronecker M: pattern split (entries outside the element-coupling pattern
                # apart), else the general line kernels
                from .kronptap import default_groups, ptap_factored
                groups = default_groups(kx.d, max(s1.p for s1 in kx.basis.splines))
                return ptap_factored(kx, Aij, (0, nz), (0, nz), (0, kz), None, 1.0, groups)
            if not scalar:
                scalar["M"] = self.M.block(0, nfe, 0, ncp)
                scalar["MT"] = scalar["M"].transpose()
                scalar["elem"] = self._element_plan_for(scalar["M"])
            if scalar["elem"] is not None and not Aij.is_loose():
                # (every block of an assembled matrix on the mixed space couples nodes of common cells of the scalar mesh)
                Kij = scalar["elem"].ptap(Aij, None, 1.0)
                if Kij is not None:
                    return Kij
            return _dev.ptap_numeric(_dev.ptap_symbolic(Aij, scalar["M"], scalar["MT"]), Aij, scalar["M"], scalar["MT"])

        blocks = []
        for i in range(nF):
            row = []
            for j in range(nF):
                Aij = A.block(i * nfe, (i + 1) * nfe, j * nfe, (j + 1) * nfe)
                Kij = None
                if Aij.nnz == 0:
                    # fields i and j are not coupled by this form: no entries in this block of the product either
                    import scipy.sparse as _sp
                    Kij = DeviceCSR.from_scipy(_sp.csr_matrix((ncp, ncp)))
                elif plan is not None:
                    pieces = []
                    for z0 in range(0, nz, step):
                        pc = plan.planes(Aij, 0, z0, min(nz, z0 + step))
                        if pc is None:
                            pieces = None
                            break
                        pieces.append(pc)
                    if pieces is not None:
                        Kij = plan.zstage(pieces, 0, kz)
                    del pieces
                if Kij is None:
                    Kij = general(Aij)
                row.append(Kij)
                del Aij
            blocks.append(row)
        K = _dev.csr_from_blocks(blocks)
        del blocks
        if zd is not None and len(zd):
            K.zero_rows_cols(numpy.asarray(zd, dtype=numpy.int32), diag)
        return K

    def _extract_matrix_by_field_list(self, A, zd, diag):
        """M^T A M for fields on DIFFERENT tensor bases (M = diag(M_f)): block (f, g) = M_f^T A_fg M_g by the line walks with
        separate row- and column-side weights where the pair qualifies (all fields on one Q_P node grid, degrees <= 3 in 3-D:
        ``TensorPtAP.for_pair``, <= 4 in 2-D: ``TensorPtAP2D.for_pair``; csrc/tg_tensor_body.h), by the general kernels on the
        scalar operands otherwise; the
        blocks are put together and MatZeroRowsColumns is applied to the whole (tIGAr/common.py:1194-1200).  None when A
        is not a matrix on this mixed space."""
        from .tensorptap import TensorPtAP
        kxs = self._kron_fields
        nF = self.nFields
        nfe = [int(numpy.prod(kx.nfe, dtype=numpy.int64)) for kx in kxs]
        ncp = [int(numpy.prod(kx.ncp, dtype=numpy.int64)) for kx in kxs]
        fo, co = numpy.concatenate([[0], numpy.cumsum(nfe)]), numpy.concatenate([[0], numpy.cumsum(ncp)])
        if A.shape != (int(fo[-1]), int(fo[-1])):
            return None
        scalar = {}

        def general(f, g, Aij):
            import scipy.sparse as _sp
            for q in (f, g):
                if q not in scalar:
                    Mq = self.M.block(int(fo[q]), int(fo[q + 1]), int(co[q]), int(co[q + 1]))
                    scalar[q] = (Mq, Mq.transpose())
            if f == g:
                return _dev.ptap_numeric(_dev.ptap_symbolic(Aij, scalar[f][0], scalar[f][1]), Aij, scalar[f][0], scalar[f][1])
            # the general kernels form P^T A P with ONE operator: block (0, 1) of the product on the two-field space
            # diag(M_f, M_g) with A_fg as its only non-zero block
            def zero(r, cc):
                return DeviceCSR.from_scipy(_sp.csr_matrix((int(r), int(cc))))
            Mp = _dev.csr_from_blocks([[scalar[f][0], zero(nfe[f], ncp[g])], [zero(nfe[g], ncp[f]), scalar[g][0]]])
            Ap = _dev.csr_from_blocks([[zero(nfe[f], nfe[f]), Aij], [zero(nfe[g], nfe[f]), zero(nfe[g], nfe[g])]])
            MpT = Mp.transpose()
            Kp = _dev.ptap_numeric(_dev.ptap_symbolic(Ap, Mp, MpT), Ap, Mp, MpT)
            return Kp.block(0, ncp[f], ncp[f], ncp[f] + ncp[g])

        blocks = []
        for f in range(nF):
            row = []
            for g in range(nF):
                Aij = A.block(int(fo[f]), int(fo[f + 1]), int(fo[g]), int(fo[g + 1]))
                Kij = None
                if Aij.nnz == 0:
                    import scipy.sparse as _sp
                    Kij = DeviceCSR.from_scipy(_sp.csr_matrix((ncp[f], ncp[g])))
                else:
                    plan = TensorPtAP.for_pair(kxs[f], kxs[g]) if (kxs[f].d == 3 and not Aij.is_loose()) else None
                    if plan is not None:
                        nz, kz = int(kxs[f].nfe[-1]), int(kxs[f].ncp[-1])
                        step = max(int(kxs[f].grid.degree), min(nz, int(2.0e10 // max(1.0, 12.0 * 2.5 * Aij.nnz / nz))))
                        pieces = []
                        for z0 in range(0, nz, step):
                            pc = plan.planes(Aij, 0, z0, min(nz, z0 + step))
                            if pc is None:
                                pieces = None
                                break
                            pieces.append(pc)
                        if pieces is not None:
                            Kij = plan.zstage(pieces, 0, kz)
                        del pieces
                    if Kij is None and kxs[f].d == 2 and not Aij.is_loose():
                        # 2-D compatible splines (demos/taylor-green/taylor-green-2d.py): the block in two walks
                        from .tensorptap import TensorPtAP2D
                        plan2 = TensorPtAP2D.for_pair(kxs[f], kxs[g])
                        if plan2 is not None:
                            Kij = plan2.ptap(Aij)
                    if Kij is None:
                        Kij = general(f, g, Aij)
                row.append(Kij)
                del Aij
            blocks.append(row)
        K = _dev.csr_from_blocks(blocks)
        del blocks
        if zd is not None and len(zd):
            K.zero_rows_cols(numpy.asarray(zd, dtype=numpy.int32), diag)
        return K

    def assembleMatrix(self, form, applyBCs=True, diag=1):
        """tIGAr/common.py:1206-1220.  When the assembled FE matrix cannot be resident next to K (implicit
        M: cfg3's A is 684 GB) or the patch is spread over several ranks, the form is asked for row blocks
        (``assemble_matrix(V, row0, row1)``) as the z-slab pipeline consumes them."""
        if not self.__dict__.get("_in_system"):
            self._new_ghost_epoch()
        if hasattr(form, "assemble_matrix"):
            if self._implicit() or self._distributed():
                from .implicit import LazyFEMatrix
                n = self.V.dim()
                fac = None
                if hasattr(form, "factors") and getattr(form, "geometry", None) is None and self.nFields == 1:
                    fac = form.factors(self.V)          # Kronecker sum of 1-D matrices: may be fused into the PtAP
                A = LazyFEMatrix(lambda r0, r1: form.assemble_matrix(self.V, r0, r1), (n, n), kron_factors=fac)
                if self.nFields > 1 and hasattr(form, "assemble_block"):
                    # forms on a mixed space hand out their field blocks (rows of one block, columns of one field)
                    A.block_rows = lambda f, g, r0, r1: form.assemble_block(self.V, f, g, r0, r1)
                    A.block_factors = form.block_factors(self.V) if hasattr(form, "block_factors") else None
            else:
                A = form.assemble_matrix(self.V)
        else:
            A = form
        K = self.extractMatrix(A, applyBCs=applyBCs, diag=diag)
        proof = None
        if getattr(form, "symmetric", False) is True and isinstance(K, DeviceCSR) and self.nFields == 1 and \
                hasattr(form, "factors") and getattr(form, "geometry", None) is None:
            proof = getattr(A, "kron_factors", None) or form.factors(self.V)      # (what A was formed from, either way)
        if proof is not None and _kron_sum_is_symmetric(proof):
            # PROVED symmetric: A is a Kronecker sum whose terms are (checked here, on the 1-D matrices) symmetric or come with
            # their transposes, so M^T A M with MatZeroRowsColumns is symmetric to rounding.  Then the solver builds its
            # half-storage copy without comparing it with the CSR product (TG_KSP_SYMMETRIC).  Every other K -- mapped forms,
            # matrices handed in, subclasses of the forms -- is compared once (the result is kept on the matrix).
            K.symmetric_by_construction = True
        return K

    def assembleLinearSystem(self, lhsForm, rhsForm, applyBCs=True):
        # one ghost update of the rank-local functions the forms read serves both assemblies (Function.ghosted)
        self._new_ghost_epoch()
        self._in_system = True
        try:
            return (self.assembleMatrix(lhsForm, applyBCs), self.assembleVector(rhsForm, applyBCs))
        finally:
            self._in_system = False

    # -- a-12
    def solveLinearSystem(self, MTAM, MTb, u):
        """Solve ``MTAM*U = MTb`` and store ``M*U`` in the FE function ``u``; returns ``U``
        (tIGAr/common.py:1236-1263).  With ``linearSolver == None`` the reference calls dolfin's direct LU;
        here: the banded LU of csrc/tg_lu.hip while its band storage fits (``_DefaultSolver``), Jacobi-GMRES at
        tight tolerance beyond, with a message that says so."""
        MTU = DeviceVector(MTAM.shape[0])          # (local rows of MTAM: all of them on one rank)
        solver = self.linearSolver if self.linearSolver is not None else _default_linear_solver()
        if getattr(solver, "parameters", {}).get("nonzero_initial_guess", False):
            # the reference sizes AND seeds MTU = M^T u.vector() (tIGAr/common.py:1250-1254): a guess set in u is used.
            # Distributed or several implicit fields: through the slab engine, which gives the rank's rows of M^T u in
            # the numbering K and M^T b carry
            if self._distributed() or (self._implicit() and self.nFields > 1):
                MTU = self._initial_guess_through_slabs(u)
            else:
                self.M.mult_transpose(_as_device_vector(u), MTU)
        if self._distributed() and getattr(solver, "comm", False) is None:
            solver.comm = self.comm.device()
        solver.solve(MTAM, MTU, MTb)
        if self._distributed() or (self._implicit() and self.nFields > 1):
            # u = M U on the FE rows this rank owns, U with its halo (tIGAr/common.py:1259-1261:
            # M*MTU followed by the ghost update)
            u_loc = self._slab_path().prolong(MTU)
            u._vec, u.local_range = u_loc, self._slab_path().mine["u_rows"]
        else:
            self.M.mult(MTU, _as_device_vector(u))
        if hasattr(u, "invalidate_ghosts"):
            u.invalidate_ghosts()              # (new coefficients: ghost rows fetched before are stale)
        return MTU

    # -- rank-local FE functions (tIGAr/common.py:1304-1348 runs on distributed PETSc vectors with ghost updates)
    def _register_ghosted(self, f):
        """``f`` is a rank-local Function whose ghost rows this spline fetches (weak reference, in registration order)"""
        import weakref
        f._ghoster = self.ghostedVector
        regs = self.__dict__.setdefault("_ghost_functions", [])
        regs[:] = [r for r in regs if r() is not None]
        if not any(r() is f for r in regs):
            regs.append(weakref.ref(f))

    def _new_ghost_epoch(self):
        """A new assembly: the ghost rows of every registered rank-local function are fetched NOW, by every rank -- also one
        whose slab holds no row block and whose forms therefore never ask (its neighbours would wait for it until the
        transport's time-out) -- and kept for the assembly (``Function.ghosted``)."""
        self._ghost_epoch = self.__dict__.get("_ghost_epoch", 0) + 1
        if self._distributed():
            for r in list(self.__dict__.get("_ghost_functions", [])):
                f = r()
                if f is not None and f.local_range is not None:
                    f.ghosted()

    def localFunction(self, u=None):
        """A Function that holds the FE rows this rank owns (``localFERange()``): a fresh one, or the rows of ``u`` cut out
        of a replicated full-length Function / vector.  The form objects read it through ``Function.ghosted()``."""
        rng = self.localFERange()
        f = Function(self.V, rng if self._distributed() else None)
        self._register_ghosted(f)
        if u is not None:
            src = _as_device_vector(u)
            if src.size() == f.vector().size():
                _dev.vec_copy_range(f.vector(), 0, src, 0, src.size())
            else:
                _dev.vec_copy_range(f.vector(), 0, src, int(rng[0]), int(rng[1]) - int(rng[0]))
        return f

    def gatherFunction(self, u):
        """The replicated, full-length Function of a rank-local one: every rank gets all FE rows (summed through the
        transport in rank order, the same bits everywhere).  After ``solveLinearSystem`` /
        ``solveNonlinearVariationalProblem`` on several ranks ``u.vector()`` holds THIS rank's FE rows only
        (``u.local_range``; the reference's distributed PETSc vector, tIGAr/common.py:1259-1261, 1343) -- code written for
        one rank that reads the whole solution (error norms, output) calls this first.  One rank, or a function that is
        already replicated: ``u`` itself."""
        vec = _as_device_vector(u)
        n = self.V.dim()
        if not self._distributed() or getattr(u, "local_range", None) is None or vec.size() == n:
            return u
        rng = u.local_range
        pieces = rng if (len(rng) and isinstance(rng[0], (tuple, list))) else [rng]
        host, full, pos = vec.get_local(), numpy.zeros(n), 0
        for (a, b) in pieces:
            a, b = int(a), int(b)
            full[a:b] = host[pos:pos + b - a]
            pos += b - a
        self.comm.transport().allreduce_sum(full)
        return Function(self.V, None, DeviceVector(data=full))

    def ghostedVector(self, u):
        """Full-length vector with the rank-local FE function ``u`` in place and the ghost rows the rank's forms read --
        the FE rows its rows of A couple to beyond its own, ``m_rows`` of its slab -- fetched from the z-neighbours
        (host-staged through the transport: one exchange per assembly, as dolfin's ghost update after ``u.assign``).
        Rows outside the window are zero."""
        vec = _as_device_vector(u)
        n = self.V.dim()
        if not self._distributed() or vec.size() == n:
            return vec
        if self.nFields != 1:
            raise NotImplementedError("ghost update of rank-local FE functions: one field")
        slab = self._slab_path()
        lay, world, rank = slab.layout, slab.world, slab.rank
        from .dist import split_range
        ranges = split_range(lay.ncp, world)
        own = [lay.slab(a, b)["u_rows"] for (a, b) in ranges]
        need = [lay.slab(a, b)["m_rows"] for (a, b) in ranges]
        # (a rank of a thin slab may own NO FE rows -- the layout reports (0, 0) -- but its forms still read a window: place
        #  the empty range where it belongs in the chain, at the end of the rows owned below it)
        pos = 0
        for r_ in range(world):
            if own[r_][1] <= own[r_][0]:
                own[r_] = (pos, pos)
            pos = own[r_][1]
        ua, ub = own[rank]
        if vec.size() != ub - ua:
            raise ValueError("ghostedVector: the function does not hold this rank's FE rows")
        host = vec.get_local()
        full = numpy.zeros(n)
        full[ua:ub] = host
        tr = self.comm.transport()
        # The transport links z-neighbours only, and a window may reach past the neighbour (slabs thinner than the
        # coupling), so the ghost rows travel in two sweeps along the chain of ranks.  The windows move upwards with the
        # rank, hence need[r-1] contains every row rank r wants from below (and need[r+1] every row it wants from above):
        # upwards, rank r first receives its lower ghost rows from r-1, then hands r+1 what that rank wants from below --
        # its own rows and, if the window is that long, rows it has just received; downwards likewise.  Which rows go
        # where follows from the layout alone.
        empty = numpy.zeros(0)
        if rank > 0 and ua > need[rank][0]:
            recv = numpy.zeros(ua - need[rank][0])
            tr.sendrecv(rank - 1, empty, recv, tag=71)
            full[need[rank][0]:ua] = recv
        if rank + 1 < world and own[rank + 1][0] > need[rank + 1][0]:
            tr.sendrecv(rank + 1, numpy.ascontiguousarray(full[need[rank + 1][0]:own[rank + 1][0]]), numpy.zeros(0), tag=71)
        if rank + 1 < world and need[rank][1] > ub:
            recv = numpy.zeros(need[rank][1] - ub)
            tr.sendrecv(rank + 1, empty, recv, tag=72)
            full[ub:need[rank][1]] = recv
        if rank > 0 and need[rank - 1][1] > own[rank - 1][1]:
            tr.sendrecv(rank - 1, numpy.ascontiguousarray(full[own[rank - 1][1]:need[rank - 1][1]]), numpy.zeros(0), tag=72)
        return DeviceVector(data=full)

    def globalNorm(self, v, kind="l2"):
        """Norm of an IGA vector over all ranks (PETSc VecNorm on the distributed M^T b, tIGAr/common.py:1330): the local
        sums go through the device communicator's all-reduce -- every rank gets the same bits."""
        if not self._distributed():
            return v.norm(kind)
        if kind != "l2":
            raise ValueError("globalNorm: l2")
        return float(numpy.sqrt(self.comm.device().allreduce_sum(numpy.array([v.inner(v)]))[0]))

    def _initial_guess_through_slabs(self, u):
        """The rank's rows of M^T u for a guess set in ``u``: a replicated full-length FE vector, or a Function that holds
        the rank's own FE rows as solveLinearSystem leaves it -- then with the ghost rows of the z-neighbours, so that the
        dofs next to a slab boundary get every contribution (the reference's MatMultTranspose on distributed vectors)"""
        vec = _as_device_vector(u)
        n = self.V.dim()
        if vec.size() == n:
            full = vec
        elif self.nFields == 1:
            full = self.ghostedVector(u)
        else:
            # several fields: the rank's own rows of every field in place (ghost rows count as zero: a guess, not a result)
            full = DeviceVector(n)
            off = 0
            for (a, b) in list(getattr(u, "local_range", None) or []):
                _dev.vec_copy_range(full, int(a), vec, off, int(b) - int(a))
                off += int(b) - int(a)

        def rows(r0, r1):
            piece = DeviceVector(r1 - r0)
            _dev.vec_copy_range(piece, 0, full, r0, r1 - r0)
            return piece
        return self._slab_path().assemble_vector(rows, None)

    def solveLinearVariationalProblem(self, residualForm, u, applyBCs=True):
        """``residualForm`` must be an ``Equation``-like object with ``.lhs`` / ``.rhs``
        forms (tIGAr/common.py:1266-1290)."""
        MTAM, MTb = self.assembleLinearSystem(residualForm.lhs, residualForm.rhs, applyBCs)
        return self.solveLinearSystem(MTAM, MTb, u)

    def setSolverOptions(self, maxIters=20, relativeTolerance=1e-5, linearSolver=None):
        """tIGAr/common.py:1292-1302."""
        self.maxIters = maxIters
        self.relativeTolerance = relativeTolerance
        self.linearSolver = linearSolver

    def FEtoIGA(self, u):
        """IGA dofs from the FE coefficients of ``u`` by the pseudo-inverse problem
        (M^T M) x = M^T u (tIGAr/common.py:968-994); uses ``self.linearSolver`` if set."""
        uv = _as_device_vector(u)
        MTtemp = self.extractVector(uv, applyBCs=False)
        ident = DeviceCSR.from_scipy(_scipy_identity(self.M.shape[0]))
        MTM = self.extractMatrix(ident, applyBCs=False)
        x = DeviceVector(self.M.shape[1])
        solver = self.linearSolver if self.linearSolver is not None else _default_linear_solver("cg")
        solver.solve(MTM, x, MTtemp)
        return x

    def solveNonlinearVariationalProblem(self, residualForm, J, u, referenceError=None, igaDoFs=None):
        """Newton iteration of tIGAr/common.py:1304-1348, same control flow: assemble
        (M^T J M, M^T R) at the current ``u``, stop when ||M^T R|| / reference < relativeTolerance
        (reference = first norm unless given), else solve for the increment and ``u <- u - du``
        (``igaDoFs -= increment`` when IGA dofs are passed; they also seed ``u = M*igaDoFs``).
        ``residualForm`` / ``J`` are form objects whose ``assemble_vector`` / ``assemble_matrix``
        read the current state of ``u`` (as UFL forms reference their coefficient).  Prints the
        reference's progress line; non-convergence raises instead of the reference's exit().
        Returns the list of relative norms."""
        returningDoFs = igaDoFs is not None
        dist = self._distributed()
        if dist:
            # several ranks: u, du and igaDoFs are rank-local (FE rows localFERange(), dofs localDofRange()), the norm is
            # global, the forms read u through Function.ghosted() -- the reference's loop on distributed PETSc vectors.
            # NOTE: a replicated ``u`` handed in becomes rank-local IN PLACE (u.vector() = this rank's rows, u.local_range
            # set); ``gatherFunction(u)`` returns the full-length function again
            if u.local_range is None:
                loc = self.localFunction(u)          # a replicated Function: every rank keeps its rows
                u._vec, u.local_range = loc.vector(), loc.local_range
            self._register_ghosted(u)
            if returningDoFs:
                u._vec = self._slab_path().prolong(igaDoFs)
        uv = _as_device_vector(u)
        if returningDoFs and not dist:
            self.M.mult(igaDoFs, uv)
        history = []
        converged = False
        for i in range(0, self.maxIters):
            MTAM, MTb = self.assembleLinearSystem(J, residualForm)
            currentNorm = self.globalNorm(MTb, "l2")
            if i == 0 and referenceError is None:
                referenceError = currentNorm
            relativeNorm = currentNorm / referenceError
            history.append(relativeNorm)
            if mpirank == 0:
                print("Solver iteration: " + str(i) + " , Relative norm: " + str(relativeNorm))
                sys.stdout.flush()
            if relativeNorm < self.relativeTolerance:
                converged = True
                break
            du = Function(self.V, self.localFERange() if dist else None)
            igaIncrement = self.solveLinearSystem(MTAM, MTb, du)
            uv.axpy(-1.0, du.vector())
            if hasattr(u, "invalidate_ghosts"):
                u.invalidate_ghosts()
            if returningDoFs:
                igaDoFs.axpy(-1.0, igaIncrement)
        if not converged:
            raise RuntimeError("Nonlinear solver failed to converge.")
        return history



def _scipy_zero(shape):
    import scipy.sparse as _sp
    return _sp.csr_matrix((int(shape[0]), int(shape[1])))


def _scipy_identity(n):
    import scipy.sparse as _sp
    return _sp.identity(n, format="csr")


class NewtonSolver(object):
    """Minimal stand-in for dolfin ``NewtonSolver`` [ext] driving an ``ExtractedNonlinearProblem``:
    ``parameters`` relative_tolerance / absolute_tolerance / maximum_iterations /
    relaxation_parameter; linear solves through ``linear_solver`` (``.solve(A,x,b)``) or the
    spline's."""

    def __init__(self, linear_solver=None):
        self.linear_solver = linear_solver
        self.parameters = {"relative_tolerance": 1e-9, "absolute_tolerance": 1e-10,
                           "maximum_iterations": 50, "relaxation_parameter": 1.0,
                           "error_on_nonconvergence": True}
        self.last = {}

    def solve(self, problem, x):
        prm = self.parameters
        r0 = None
        for it in range(int(prm["maximum_iterations"]) + 1):
            problem.form(None, None, None, x)
            b = problem.F(None, x)
            rn = b.norm("l2")
            r0 = rn if r0 is None else r0
            self.last = {"iterations": it, "residual": rn, "relative": rn / r0 if r0 > 0 else 0.0}
            if rn < prm["absolute_tolerance"] or (r0 > 0 and rn / r0 < prm["relative_tolerance"]):
                return it, True
            if it == int(prm["maximum_iterations"]):
                break
            A = problem.J(None, x)
            dx = DeviceVector(x.size())
            ls = self.linear_solver or problem.spline.linearSolver or _default_linear_solver()
            ls.solve(A, dx, b)
            x.axpy(-float(prm["relaxation_parameter"]), dx)
        if prm["error_on_nonconvergence"]:
            raise RuntimeError("Newton solver did not converge")
        return int(prm["maximum_iterations"]), False


class ExtractedNonlinearProblem(object):
    """Nonlinear problem posed on an extracted spline for external Newton/SNES-type solvers
    (tIGAr/common.py:504-545): ``form`` pushes the IGA dofs into the FE solution (M*x), ``F`` / ``J``
    return the extracted residual / tangent."""

    def __init__(self, spline, residual, tangent, solution, **kwargs):
        self.spline, self.residual, self.tangent, self.solution = spline, residual, tangent, solution

    def form(self, A, P, B, x):
        self.spline.M.mult(x, self.solution.vector())

    def F(self, b, x):
        return self.spline.assembleVector(self.residual)

    def J(self, A, x):
        return self.spline.assembleMatrix(self.tangent)


class ExtractedNonlinearSolver(object):
    """tIGAr/common.py:547-584: initial IGA dofs by ``FEtoIGA`` of the current solution, solve, then
    the FE representation of the result is stored in ``problem.solution``."""

    def __init__(self, problem, solver):
        self.problem, self.solver = problem, solver

    def solve(self):
        tempVec = self.problem.spline.FEtoIGA(self.problem.solution)
        self.solver.solve(self.problem, tempVec)
        self.problem.spline.M.mult(tempVec, self.problem.solution.vector())
        return tempVec
