import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
dt = torch.bfloat16
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (25088, 512, 2048), (100352, 512, 2048), (100352, 1024, 512), (25088, 2048, 512)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt); b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    kw = dict(M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N)
    for _ in range(3): ops.gemm(a, b, c, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(a, b, c, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"M={M} N={N} K={K}: {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")
