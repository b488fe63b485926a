"""What the t2s stage-1 core would cost on (s, t)-ordered scores / values: the s2t code path with T and S swapped."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bist_amd import ops
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, S, Lq, h, dk = 16, 49, 20, 8, 64
d = h * dk
for T in (32, 128):
    sc = torch.randn(B, Lq * h, T * S, device="cuda")
    v = torch.randn(B, T, S, d, device="cuda").bfloat16()
    tm = torch.ones(B, 1, T, dtype=torch.bool, device="cuda")
    a = timeit(lambda: ops.st_stage1_pv(sc, v, tm, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk, direction=0))
    vp = v.permute(0, 2, 1, 3).contiguous()
    b = timeit(lambda: ops.st_stage1_pv(sc, vp, None, B=B, T=S, S=T, Lq=Lq, h=h, dk=dk, direction=1))
    print(f"T={T}: t2s as it is {a:.1f} us; on permuted tensors (s2t code path, no mask) {b:.1f} us")
