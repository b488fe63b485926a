"""bist_amd -- MI355X-native (gfx950) implementation of BiST's bi-directional spatio-temporal
attention hot path: Python host code over hand-written HIP kernels behind a C ABI
(include/bist_hip.h).  See DESIGN.md."""
from . import _lib  # noqa: F401  (raises if libbist_hip.so is missing: there is no fallback)

__version__ = "0.1.0"
