"""Build-time check of st1_fused.hip (development / CI aid): its inline-asm loads (ds_read_b128 / global_load_dwordx4 with
counted waits) write their destination registers asynchronously, which the compiler does not know -- a register spilled or
re-assigned while such a load is in flight would be corrupted.  The kernel is written to fit its 256 registers; this script
compiles it to ISA and reports, per st1_fused kernel, the scratch traffic and whether any spilled register is ever the destination
of an inline-asm load (must be none).  usage: python scripts/check_spills.py [path to st1_fused.hip]"""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bist_amd", "csrc", "st1_fused.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = False
name, in_asm, fly_g, fly_l, spilled, hits = None, False, set(), set(), 0, []
for ln, line in enumerate(text + ["END:"]):
    m = re.match(r"^(_ZN\S*st1_fused_kernel\S*|END):", line)
    if m or (line.startswith("_Z") and line.endswith(":") and name):
        if name:
            print(f"{name[:60]}: {spilled} spill stores; spills of registers with an inline-asm load in flight: {hits or 'none'}")
            bad |= bool(hits)
        name = m.group(1) if m and m.group(1) != "END" else None
        fly_g, fly_l, spilled, hits = set(), set(), 0, []
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
    elif t.startswith(";;#ASMEND"):
        in_asm = False
    elif in_asm and t.startswith("ds_read_b128"):
        fly_l |= regs(t.split()[1].rstrip(","))
    elif in_asm and t.startswith("global_load_dwordx4"):
        fly_g |= regs(t.split()[1].rstrip(","))
    elif t.startswith("s_waitcnt"):
        # a full drain retires everything of that counter; the counted waits of the streams are followed by their tie() asm
        # (an empty ASMSTART/ASMEND pair), so registers stay "in flight" here until a full drain: conservative
        if "vmcnt(0)" in t:
            fly_g = set()
        if "lgkmcnt(0)" in t:
            fly_l = set()
    elif t.startswith("scratch_store"):
        spilled += 1
        r = regs(t.split()[2].rstrip(","))
        if r & (fly_g | fly_l):
            hits.append((ln, t.split(";")[0].strip()))
sys.exit(1 if bad else 0)
