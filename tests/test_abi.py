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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of the by-reference structs as a C compiler sees include/bist_hip.h against the ctypes mirrors."""
    import os
    import subprocess
    from bist_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"BistGemm": ["a_rs", "alpha", "drop_seed", "workspace", "hint", "ln_gain", "ln_ld", "ln_eps", "ln_mode", "bias_bs1"],
               "BistDecLayer": ["Wqkv", "Wo", "cmask", "W1", "Lk", "LkP"], "BistDrop": ["seed", "ctr"],
               "BistLnSet": ["a", "y"], "BistLnBwdSet": ["dx", "db", "dz", "drop_row0"]}
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "bist_hip.h"', "int main(void) {"]
    for st, fields in structs.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        src += [f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fields]
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(exe)], check=True)
    seen = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for st, fields in structs.items():
        cls = getattr(_lib, st)
        assert ctypes.sizeof(cls) == int(seen[st]), st
        for f in fields:
            assert getattr(cls, f).offset == int(seen[f"{st}.{f}"]), (st, f)


def test_ops_refuse_cpu_tensors():
    import torch
    from bist_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))


def test_reference_pickle_resolves_to_this_build_on_cpu(golden_dir):
    """generate.py:93 ``torch.load`` of a whole-module pickle written by the reference (class paths ``model.*``): with
    INTEGRATION.md's module mapping it becomes an instance of this build's MTN with every submodule present (no compute)."""
    import os
    import sys
    import torch
    import bist_amd.model as m
    names = ("", ".mtn", ".modules", ".encoder", ".decoder", ".generator", ".label_smoothing", ".optimize", ".decode")
    saved = {("model" + n): sys.modules.get("model" + n) for n in names}
    try:
        for n in names:
            sys.modules["model" + n] = sys.modules["bist_amd.model" + n] if n else m
        obj = torch.load(os.path.join(golden_dir, "g7_reference_module.pth.tar"), weights_only=False)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert type(obj) is m.mtn.MTN
    assert obj.generator.vocab_gen is obj.query_embed[0].lut.weight          # the shared embedding survives the pickle
    sub = obj.mutlimodal_decoder.v_layers[0].sublayer[0]
    assert sub.p == sub.dropout.p and obj.mutlimodal_decoder.v_layers[0].attn[0].keep_attn is False


def test_no_spill_of_a_register_with_an_inline_asm_load_in_flight():
    """The fused stage-1 kernel's inline-asm loads write registers the compiler does not track (counted waits): the ISA guard that
    __graft_entry__.build() runs must pass for the sources as they are (a compiler update that starts spilling such a register
    would corrupt results silently between GPU runs)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("/opt/rocm/bin/hipcc") is None:
        pytest.skip("hipcc is not installed here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spills.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("inline-asm load in flight: none") >= 5, r.stdout
