"""Per-workgroup time of the 256-tile GEMM with the A operand resident in the Infinity Cache vs streamed from HBM
(development aid): same N, K, one round of workgroups in both cases."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
dt = torch.bfloat16
N, K = 512, 2048
for M, note in [(25088, "A 103 MB: streamed from HBM"), (25088 // 4, "A 26 MB: Infinity-Cache resident after warm-up"), (40960, "160 row tiles")]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt); b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    kw = dict(M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N)
    for _ in range(5): ops.gemm(a, b, c, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(a, b, c, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M:6d} ({note}): {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s  plan={ops._lib.lib.bist_gemm_is_fast(ops.gemm_desc(a,b,c,**kw))}")
