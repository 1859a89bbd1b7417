"""
Div- and curl-conforming (RT / N type) B-spline fields: the construction half of
``tIGAr/compatibleSplines.py`` (Buffa et al., https://epubs.siam.org/doi/10.1137/100786708) --
``generateFieldsCompat`` (:21-66) and ``BSplineCompat`` (:69-101).  The resulting multi-field spline
(per-field, per-direction degrees) is extracted by the same HIP kernels as every other
``AbstractMultiFieldSpline``.  The UFL-level half of the reference module (``iteratedDivFreeSolve``,
``ExtractedBSplineRT/N``, ``pushforward*``, ``div``, ``curl``) needs FEniCS and is not provided.
"""
import numpy

from .common import AbstractMultiFieldSpline
from .BSplines import BSpline


def _raised_directions(RTorN, nvar, field):
    """Mask over the parametric directions: True where component ``field`` is one degree higher.
    RT: along its own direction; N: in every other direction."""
    own = numpy.arange(nvar) == field
    if RTorN == "RT":
        return own
    if RTorN == "N":
        return ~own
    return numpy.zeros(nvar, dtype=bool)          # (the reference raises nothing for other tags)


def _knot_vector(breakpoints, degree, periodic):
    """Breakpoints with the end knots repeated to multiplicity degree+1 (open vector); periodic
    directions keep the bare breakpoints (tIGAr/compatibleSplines.py:54-61)."""
    u = numpy.asarray(breakpoints, dtype=numpy.float64)
    if periodic:
        return u.copy()
    return numpy.r_[numpy.full(degree, u[0]), u, numpy.full(degree, u[-1])]


def generateFieldsCompat(controlMesh, RTorN, degrees, periodicities=None):
    """Scalar ``BSpline`` bases of the components of an RT- or N-type compatible discretisation on
    the breakpoints of the control mesh's scalar spline (tIGAr/compatibleSplines.py:21-66).
    ``degrees`` are the base degrees per direction, ``periodicities`` an optional list of flags."""
    base = controlMesh.getScalarSpline()
    nvar = len(degrees)
    periodic = [bool(periodicities[k]) if periodicities is not None else False for k in range(nvar)]
    out = []
    for field in range(nvar):
        q = numpy.asarray(degrees, dtype=int) + _raised_directions(RTorN, nvar, field)
        kvecs = [_knot_vector(base.splines[k].uniqueKnots, int(q[k]), periodic[k]) for k in range(nvar)]
        out.append(BSpline([int(v) for v in q], kvecs, base.useRectangularElements()))
    return out


class BSplineCompat(AbstractMultiFieldSpline):
    """Extraction generator of a compatible spline of type RT or N with no further fields
    (tIGAr/compatibleSplines.py:69-101): ``BSplineCompat(controlMesh, "RT"|"N", degrees[,
    periodicities])``."""

    def customSetup(self, args):
        self.controlMesh, self.RTorN, self.degrees = args[0], args[1], args[2]
        self.periodicities = args[3] if len(args) > 3 else None
        self.fields = generateFieldsCompat(self.controlMesh, self.RTorN, self.degrees, self.periodicities)

    def getControlMesh(self):
        return self.controlMesh

    def getNFields(self):
        return len(self.fields)

    def getFieldSpline(self, field):
        return self.fields[field]
