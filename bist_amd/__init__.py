"""bist_amd -- MI355X-native (gfx950) implementation of BiST's bi-directional spatio-temporal
attention hot path: Python host code over hand-written HIP kernels behind a C ABI
(include/bist_hip.h).  See DESIGN.md."""
from . import _lib  # noqa: F401  (raises if libbist_hip.so is missing: there is no fallback)

__version__ = "0.1.0"

# The split-graph executor (bist_amd/graphsplit.py) launches one linear hipGraph per stream and needs every stream to stay on a hardware
# queue of its own: the runtime's dynamic stream -> queue mapping moves a stream's later packets onto another stream's queue (measured:
# profiles/r04_split_queue_aliasing.txt), and four queues are too few beside the framework's own streams; and the runtime's own graph executor must treat every graph it is
# handed as ONE queue (its pre-built packet path, 0.3 us of host time per node).  The switches are read by
# the HIP runtime once, at its initialisation -- so they are set here, at import, unless the process has already initialised it
# (bist_amd.graphsplit.usable() then finds the executor's self-test failing and the trainer replays through the runtime's own executor).
import os as _os

import torch as _torch

if not _torch.cuda.is_initialized() and _os.environ.get("BIST_SPLIT_GRAPH", "1") != "0":
    _os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0")
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")      # every graph the runtime replays is a single-queue graph: its fast path
