"""matchinglib_poselib_amd -- MI355X-native (gfx950) descriptor-matching + robust-pose hot path.

A from-scratch HIP implementation of the `getMatches(...,"LINEAR")` brute-force 2-NN + ratio test and of the
RANSAC / Nister 5-point / Sampson / cheirality path of josefmaierfl/matchinglib_poselib, behind the C ABI
declared in include/mlpl_c.h (libmlpl_hip.so).  This package is the thin host-side mirror of the
reference's operator interface; all arithmetic runs in the HIP library and there is no CPU fallback.
"""
from ._lib import MlplError, Context, load_library, library_path  # noqa: F401
from .matching import DMATCH_DTYPE, getMatches, knn_hamming, knn_l2sq, ratio_compact  # noqa: F401

__all__ = [
    "MlplError",
    "Context",
    "load_library",
    "library_path",
    "DMATCH_DTYPE",
    "getMatches",
    "knn_hamming",
    "knn_l2sq",
    "ratio_compact",
]
