"""
Div- and curl-conforming (RT / N type) B-spline fields, mirroring the construction part of
``tIGAr/compatibleSplines.py`` (Buffa et al., https://epubs.siam.org/doi/10.1137/100786708):
``generateFieldsCompat`` (:21-66) and ``BSplineCompat`` (:69-101).  Extraction of the resulting
multi-field spline (different degrees per field and direction) runs through the same HIP kernels as
every other ``AbstractMultiFieldSpline``.  The UFL-level helpers of the reference module
(``iteratedDivFreeSolve``, ``ExtractedBSplineRT/N`` with ``pushforward*``, ``div``, ``curl``) need FEniCS
and are not part of this package.
"""
import copy

from numpy import array, concatenate

from .common import AbstractMultiFieldSpline
from .BSplines import BSpline


def generateFieldsCompat(controlMesh, RTorN, degrees, periodicities=None):
    """List of ``BSpline`` scalar bases for the components of an RT- or N-type compatible spline
    discretisation: field i is k-refined along (RT) or perpendicular to (N) direction i, re-using
    the unique knots of the control mesh's scalar ``BSpline``; open knot vectors unless periodic
    (tIGAr/compatibleSplines.py:21-66)."""
    nvar = len(degrees)
    useRect = controlMesh.getScalarSpline().useRectangularElements()
    fields = []
    for i in range(0, nvar):
        knotVectors = []
        scalarDegrees = []
        for j in range(0, nvar):
            degree = degrees[j]
            if ((RTorN == "RT") and (j == i)) or ((RTorN == "N") and (not j == i)):
                degree += 1
            knots = copy.copy(controlMesh.getScalarSpline().splines[j].uniqueKnots)
            if periodicities is None or (not periodicities[j]):
                for k in range(0, degree):
                    knots = concatenate((array([knots[0], ]), knots, array([knots[-1], ])))
            knotVectors += [knots, ]
            scalarDegrees += [degree, ]
        fields += [BSpline(scalarDegrees, knotVectors, useRect), ]
    return fields


class BSplineCompat(AbstractMultiFieldSpline):
    """Extraction generator for a compatible spline of type RT or N with no other fields
    (tIGAr/compatibleSplines.py:69-101): ``BSplineCompat(controlMesh, "RT"|"N", degrees[,
    periodicities])``."""

    def customSetup(self, args):
        self.controlMesh = args[0]
        self.RTorN = args[1]
        self.degrees = args[2]
        self.periodicities = args[3] if len(args) > 3 else None
        self.fields = generateFieldsCompat(self.controlMesh, self.RTorN, self.degrees,
                                           periodicities=self.periodicities)

    def getControlMesh(self):
        return self.controlMesh

    def getFieldSpline(self, field):
        return self.fields[field]

    def getNFields(self):
        return len(self.fields)
