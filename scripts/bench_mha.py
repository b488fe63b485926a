"""Small-attention kernels alone, replayed from a hipGraph (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import functional as Fn
N, Lq, Lk, h, d = 16, 20, 20, 8, 512
dt = torch.bfloat16
q = torch.randn(N, Lq, d, device="cuda").to(dt).requires_grad_(True)
kv = torch.randn(N, Lk, 2 * d, device="cuda").to(dt).requires_grad_(True)
mask = torch.ones(N, 1, Lk, dtype=torch.bool, device="cuda")
go = torch.randn(N, Lq, d, device="cuda").to(dt)
def fwd_bwd():
    o, _ = Fn.mha_packed(q, kv, None, "q_kv", mask, h, False, None)
    o.backward(go)
def fwd():
    with torch.no_grad():
        Fn.mha_packed(q, kv, None, "q_kv", mask, h, False, None)
def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tf = timeit(fwd); tfb = timeit(fwd_bwd)
print(f"mha fwd {tf:.1f} us; fwd+bwd {tfb:.1f} us -> bwd ~{tfb - tf:.1f} us (N={N} Lq={Lq} Lk={Lk} h={h} dk={d//h})")
