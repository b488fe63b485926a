"""Times bist_st_stage1_fused_fwd alone (development aid).  BIST_ST1F_DBG ablation bits: see st1_fused.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops

B = int(os.environ.get("B", "64")); T = int(os.environ.get("T", "32")); S = int(os.environ.get("S", "49")); Lq, d, h = 20, 512, 8
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda()
vft, qf, x = r(B, T, S, d), r(B, Lq * h, d, sc=d ** -0.5), r(B, Lq, d)
wv, bv, wo, bo = r(d, d, sc=d ** -0.5), r(d, sc=0.1), r(d, d, sc=d ** -0.5), r(d, sc=0.1)
wv, wo = ops.pack_frag_rows(wv), ops.pack_frag_rows(wo)
tm = torch.ones(B, T, dtype=torch.bool).cuda()
for direction in (0, 1):
    G = S if direction == 0 else T
    out = torch.empty(B, G, Lq, d, dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.st_stage1_fused(qf, vft, tm if direction == 0 else None, wv, bv, wo, bo, x, h=h, direction=direction, out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    K = T if direction == 0 else S
    fl = B * G * (4 * K * d * d + 4 * Lq * K * d + 2 * Lq * d * d)     # reference formulation: K,V proj + QK^T + PV + out-proj
    print(f"dbg={os.environ.get('BIST_ST1F_DBG', '0')} dir={direction} B={B} T={T}: {us:.1f} us  ({fl / us / 1e6:.0f} TFLOP/s of the reference formulation)")
