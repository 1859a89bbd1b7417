"""
tigar_amd -- MI355X-native implementation of tIGAr's extraction hot path
(extraction-operator build, M^T A M / M^T b, Krylov solve) behind tIGAr's
AbstractExtractionGenerator / ExtractedSpline API.  Host side is Python; all numerics run
in hand-written HIP kernels (libtigar_hip.so) reached through ctypes.
"""
__version__ = "0.1.0"
