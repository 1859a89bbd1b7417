"""
tigar_amd -- MI355X-native implementation of tIGAr's extraction hot path
(extraction-operator build, M^T A M / M^T b, Krylov solve) behind tIGAr's
AbstractExtractionGenerator / ExtractedSpline API.  Host side is Python; all numerics run
in hand-written HIP kernels (libtigar_hip.so) reached through ctypes.

``from tigar_amd import *`` / ``from tigar_amd.BSplines import *`` mirror
``from tIGAr import *`` / ``from tIGAr.BSplines import *``.
"""
__version__ = "0.1.0"
from .common import *          # noqa: F401,F403
from .common import (AbstractExtractionGenerator, AbstractCoordinateChartSpline, AbstractScalarBasis,
                     AbstractControlMesh, AbstractMultiFieldSpline, EqualOrderSpline, FieldListSpline,
                     ExtractedSpline, PETScKrylovSolver, KrylovSolver, Function, TensorFunctionSpace,
                     TensorNodeGrid, multTranspose, generateIdentityPermutation,
                     ExtractedNonlinearProblem, ExtractedNonlinearSolver, NewtonSolver)
from .NURBS import NURBSControlMesh      # noqa: E402,F401
