"""MI355X-native SIFT / SURF feature extraction and descriptor matching.

Python host side of libmi_features.so (HIP kernels behind a C ABI, see
include/mi_features.h). There is no CPU fallback: importing the bindings or
calling an operator without the built library raises.
"""
from ._lib import MifError, load_library, library_path  # noqa: F401
from .sift import SiftExtractor, SiftFeatures, extract_features  # noqa: F401
from .match import Matcher, ShardedMatcher, match_sift_features, match_ratio_test, pack_sift_rows  # noqa: F401
from .surf import SurfExtractor, IPOINT_DTYPE  # noqa: F401
from . import surf  # noqa: F401
from .video import VideoPipeline, FrameSummary  # noqa: F401
