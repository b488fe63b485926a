"""bist_st_stage1_fused_fwd vs bist_st_stage1_fused_raw_fwd, each alone, at the region's shape (development aid, round 4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
B, T, S, Lq, d, h = int(os.environ.get("B", "64")), 32, 49, 20, 512, 8
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)
xraw = torch.relu(r(B, T, S, d) * 0.8 + 0.3)
a, beta = (1 + 0.2 * torch.randn(d, device="cuda", generator=g)).to(dt), r(d, sc=0.1)
xn = ops.layernorm(xraw, a, beta, 1e-6)
qf = r(B, Lq * h, d, sc=1.5 * d ** -0.5)
wv, bv, wo, bo = r(d, d, sc=d ** -0.5), r(d, sc=0.1), r(d, d, sc=d ** -0.5), r(d, sc=0.1)
x = r(B, Lq, d)
wvf, wof = ops.pack_frag_rows(wv), ops.pack_frag_rows(wo)
wbar = wv.float().sum(1).contiguous()
km = torch.ones(B, T, dtype=torch.bool, device="cuda")
for direction in (0, 1):
    m = km if direction == 0 else None
    for name, fn in (("plain", lambda: ops.st_stage1_fused(qf, xn, m, wvf, bv, wof, bo, x, h=h, direction=direction)),
                     ("raw", lambda: ops.st_stage1_fused(qf, xraw, m, wvf, bv, wof, bo, x, h=h, direction=direction, raw=(wbar, 1e-6)))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"direction {direction} {name:5s}: {[round(t, 1) for t in ts]} us per launch", flush=True)
