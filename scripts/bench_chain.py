"""Dependent chain of small GEMMs in a hipGraph (development aid): what one small kernel costs in situ,
with hot vs cold weights and with a dependent vs independent input."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops

M, d, L = 320, 512, 128
dt = torch.bfloat16
Ws = [torch.randn(d, d, device="cuda").to(dt) * d ** -0.5 for _ in range(L)]
big = [torch.randn(64, 1024, 1024, device="cuda") for _ in range(2)]      # 512 MiB of eviction traffic
xs = [torch.randn(M, d, device="cuda").to(dt) for _ in range(L + 1)]


def run(name, fn, evict=False):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    ts = []
    for _ in range(5):
        if evict:
            big[0].copy_(big[1])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / L)
    print(f"{name:60s} {min(ts):7.2f} us/kernel (median {sorted(ts)[2]:.2f})")


def chain_cold():
    for i in range(L):
        ops.gemm(xs[i], Ws[i], xs[i + 1], M=M, N=d, K=d, a_rs=d, a_ks=1, b_rs=d, b_ks=1, ldc=d)

def chain_hot():
    for i in range(L):
        ops.gemm(xs[i % 2], Ws[0], xs[(i + 1) % 2], M=M, N=d, K=d, a_rs=d, a_ks=1, b_rs=d, b_ks=1, ldc=d)

def indep_hot():
    for i in range(L):
        ops.gemm(xs[0], Ws[0], xs[1], M=M, N=d, K=d, a_rs=d, a_ks=1, b_rs=d, b_ks=1, ldc=d)

def ln_chain():
    a = torch.ones(d, device="cuda", dtype=dt); b = torch.zeros(d, device="cuda", dtype=dt)
    def f():
        for i in range(L):
            ops.layernorm(xs[i % 2], a, b, 1e-6, out=xs[(i + 1) % 2]) if False else ops.layernorm(xs[i % 2], a, b, 1e-6)
    return f

run("independent, same operands (all hot)", indep_hot)
run("dependent chain, one weight (hot weight)", chain_hot)
run("dependent chain, 128 distinct weights (64 MiB)", chain_cold)
run("dependent chain, 128 weights, caches evicted before replay", chain_cold, evict=True)
