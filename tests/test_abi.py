"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/bist_hip.h declares, and the ctypes table binds exactly that set (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bist_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bist_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bist_amd import _lib
    names = _declared()
    assert len(names) >= 12
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in bist_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_version_and_error_string_without_gpu():
    from bist_amd import _lib
    assert _lib.lib.bist_version() >= 100
    assert isinstance(_lib.lib.bist_last_error(), bytes)


def test_gemm_struct_layout_matches_header():
    from bist_amd import _lib
    # values printed by a C program including include/bist_hip.h (sizeof, offsetof)
    assert ctypes.sizeof(_lib.BistGemm) == 256
    assert _lib.BistGemm.a_rs.offset == 56 and _lib.BistGemm.alpha.offset == 184 and _lib.BistGemm.drop_seed.offset == 216


def test_ops_refuse_cpu_tensors():
    import torch
    from bist_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))
