"""
The ``NURBS`` module of tigar_amd.  The reference's ``NURBSControlMesh`` (tIGAr/NURBS.py:13-77)
reads PetIGA files / igakit objects; igakit is an external dependency outside the hot path, so
here the control mesh is built directly from the data igakit would provide: degrees, knot
vectors and the homogeneous control net.  As in the reference, the extraction operator is that of
the *polynomial* B-spline space; the rational weights enter only through the control functions
``cpFuncs[nsd] = M_control * w`` (tIGAr/common.py:367-380).
"""
import numpy

from .common import AbstractControlMesh, USE_RECT_ELEM_DEFAULT
from .BSplines import BSpline
from . import device as _dev


class NURBSControlMesh(AbstractControlMesh):
    """NURBS geometry: ``NURBSControlMesh(degrees, kvecs, control)`` with ``control`` of shape
    (M, [N, [O,]] dim) in homogeneous form (w*x, w*y, ..., w), or any object with ``.degree``,
    ``.knots`` and ``.control`` attributes (an igakit ``NURBS``)."""

    def __init__(self, degrees_or_nurbs, kvecs=None, control=None, useRect=USE_RECT_ELEM_DEFAULT, overRefine=0):
        if kvecs is None:
            ik = degrees_or_nurbs
            degrees, kvecs, control = ik.degree, ik.knots, ik.control
        else:
            degrees = degrees_or_nurbs
        self.scalarSpline = BSpline(list(degrees), [numpy.asarray(k, dtype=numpy.float64) for k in kvecs],
                                    useRect, overRefine)
        control = numpy.asarray(control, dtype=numpy.float64)
        nvar = len(degrees)
        dim = control.shape[-1]
        # bnet[ij2dof(i,j,M), :] = control[i, j, :]   (tIGAr/NURBS.py:46-66): first index fastest.  Stored column by
        # column (Fortran order): a coordinate of all control points is one contiguous piece -- what M_control multiplies
        n = int(numpy.prod(control.shape[:-1]))
        self.bnet = numpy.empty((n, dim), order="F")
        for c in range(dim):
            self.bnet[:, c] = control[..., c].ravel(order="F")
        self._device_columns = {}
        if self.bnet.shape[0] != self.scalarSpline.getNcp():
            raise ValueError("control net has %d points, the spline space %d"
                             % (self.bnet.shape[0], self.scalarSpline.getNcp()))

    def getScalarSpline(self):
        return self.scalarSpline

    def getHomogeneousCoordinate(self, node, direction):
        return self.bnet[node, direction]

    def getHomogeneousCoordinates(self):
        return self.bnet

    def homogeneousCoordinateDeviceVector(self, direction):
        """Column ``direction`` of the control net in HBM.  The control net is an INPUT of the path, like the knot vectors:
        it is uploaded once per control mesh and stays resident (a generator built from the same mesh again -- every step
        of a Newton or time loop, every step of bench.py -- reads the resident copy; at cfg3's size the strided host
        gather + the upload of the four columns cost 0.12 s per generator).  ``bnet`` is not expected to change after
        construction; ``invalidateDeviceCopy()`` drops the resident columns if it does."""
        v = self._device_columns.get(direction)
        if v is None:
            v = _dev.DeviceVector(data=numpy.ascontiguousarray(self.bnet[:, direction]))
            self._device_columns[direction] = v
        return v

    def invalidateDeviceCopy(self):
        self._device_columns = {}

    def getNsd(self):
        return self.bnet.shape[1] - 1
