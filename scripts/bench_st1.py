"""Micro-benchmark of the stage-1 / stage-2 attention cores at the benchmark geometry (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops

B = int(os.environ.get("B", 64)); T = int(os.environ.get("T", 32)); S, Lq, h, dk = 49, 20, 8, 64
d = h * dk
sc = torch.randn(B, Lq * h, T * S, device="cuda")
v = torch.randn(B, T, S, 2 * d, device="cuda").bfloat16()
tm = torch.ones(B, 1, T, dtype=torch.bool, device="cuda")


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for direction in (0, 1):
    G = S if direction == 0 else T
    us = timeit(lambda: ops.st_stage1_pv(sc, v[..., :d], tm if direction == 0 else None, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk, direction=direction))
    mb = (sc.numel() * 4 + B * T * S * d * 2 + B * G * Lq * d * 2) / 1e6
    print(f"st1 fwd dir{direction} B={B}: {us:8.1f} us   {mb:.0f} MB -> {mb/us*1e-3*1e3:.0f} GB/s")
for G in (49, 32):
    q2f = torch.randn(B, Lq, h, d, device="cuda").bfloat16()
    y = torch.randn(B, G, Lq, d, device="cuda").bfloat16()
    us = timeit(lambda: ops.st_stage2(q2f, y, None, h=h))
    mb = (y.numel() * 2 + 2 * q2f.numel() * 2) / 1e6
    print(f"st2 fwd G={G} B={B}: {us:8.1f} us   {mb:.0f} MB -> {mb/us*1e3:.0f} GB/s")
x = torch.randn(B * T * S, d, device="cuda").bfloat16()
a_, b_ = torch.ones(d, device="cuda").bfloat16(), torch.zeros(d, device="cuda").bfloat16()
us = timeit(lambda: ops.layernorm(x, a_, b_))
print(f"layernorm rows={B*T*S}: {us:8.1f} us   {x.numel()*4/1e6:.0f} MB -> {x.numel()*4/us*1e-3:.0f} GB/s")
