"""Micro-benchmark of bist_gemm on the shapes of the hot path (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops, _lib

SHAPES = [  # (name, M, N, K, layout)  layout: NT = x.W^T, NN = dy.W, TN = dy^T.x
    ("small fwd  M=320  N=512  K=512", 320, 512, 512, "NT"),
    ("small fwd  M=320  N=512  K=2048", 320, 512, 2048, "NT"),
    ("small fwd  M=320  N=1536 K=512", 320, 1536, 512, "NT"),
    ("small fwd  M=1280 N=512  K=512", 1280, 512, 512, "NT"),
    ("small fwd  M=320  N=2048 K=512", 320, 2048, 512, "NT"),
    ("small dx   M=320  N=512  K=512", 320, 512, 512, "NN"),
    ("small dx   M=320  N=512  K=2048", 320, 512, 2048, "NN"),
    ("small dx   M=320  N=2048 K=512", 320, 2048, 512, "NN"),
    ("dW small   M=512 N=2048 K=320", 512, 2048, 320, "TN"),
    ("dW small   M=2048 N=512 K=320", 2048, 512, 320, "TN"),
    ("P0   B=16  M=25088 N=512 K=2048", 25088, 512, 2048, "NT"),
    ("P0   B=64  M=100352 N=512 K=2048", 100352, 512, 2048, "NT"),
    ("V    B=16  M=25088 N=512 K=512", 25088, 512, 512, "NT"),
    ("V14  B=64  M=100352 N=1024 K=512", 100352, 1024, 512, "NT"),
    ("oprj B=16  M=15680 N=512 K=512", 15680, 512, 512, "NT"),
    ("dX   B=16  M=25088 N=512 K=512", 25088, 512, 512, "NN"),
    ("dW   B=16  M=512 N=512 K=25088", 512, 512, 25088, "TN"),
    ("dWp0 B=16  M=512 N=2048 K=25088", 512, 2048, 25088, "TN"),
    ("dW small   M=512 N=512 K=320", 512, 512, 320, "TN"),
]
dt = torch.bfloat16
if os.environ.get('BIG'):
    SHAPES = [x for x in SHAPES if 'small' not in x[0]]
if os.environ.get('SMALL'):
    SHAPES = [x for x in SHAPES if 'small' in x[0]]
for name, M, N, K, lay in SHAPES:
    if lay == "NT":
        a, b = torch.randn(M, K, device="cuda").to(dt), torch.randn(N, K, device="cuda").to(dt)
        kw = dict(a_rs=K, a_ks=1, b_rs=K, b_ks=1)
    elif lay == "NN":
        a, b = torch.randn(M, K, device="cuda").to(dt), torch.randn(K, N, device="cuda").to(dt)
        kw = dict(a_rs=K, a_ks=1, b_rs=1, b_ks=N)
    else:
        a, b = torch.randn(K, M, device="cuda").to(dt), torch.randn(K, N, device="cuda").to(dt)
        kw = dict(a_rs=1, a_ks=M, b_rs=1, b_ks=N)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    plan = _lib.lib.bist_gemm_is_fast(ops.gemm_desc(a, b, c, M=M, N=N, K=K, ldc=N, **kw))
    for _ in range(5):
        ops.gemm(a, b, c, M=M, N=N, K=K, ldc=N, **kw)
    torch.cuda.synchronize()
    iters = 50
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm(a, b, c, M=M, N=N, K=K, ldc=N, **kw)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()          # replay from a graph: GPU time per launch, no host launch overhead
    with torch.cuda.graph(g):
        for _ in range(iters):
            ops.gemm(a, b, c, M=M, N=N, K=K, ldc=N, **kw)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    ref = ""
    if os.environ.get("VS_BLAS"):       # yardstick only: hipBLASLt through torch on the same operands
        fa = (lambda: torch.mm(a, b.t(), out=c)) if lay == "NT" else (lambda: torch.mm(a, b, out=c)) if lay == "NN" else (lambda: torch.mm(a.t(), b, out=c))
        for _ in range(5): fa()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters): fa()
        e1.record(); torch.cuda.synchronize()
        ru = e0.elapsed_time(e1) / iters * 1e3
        ref = f"   hipBLASLt {ru:8.1f} us {2.0*M*N*K/ru/1e6:8.1f} TFLOP/s"
    print(f"{name:38s} plan={plan} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s{ref}")
