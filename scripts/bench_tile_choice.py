"""128-tile vs 256-tile kernel on the mid-size K-contiguous products of a training step (development aid)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
from bist_amd._lib import lib, check

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (M, N, K) in [(15680, 512, 512), (10240, 512, 512), (15680, 160 * 1, 512), (25088, 512, 512), (25088, 1024, 512), (15680, 512, 2048)]:
    # rotate operands so that A streams from HBM
    xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(8)]
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16()
    out = {}
    for hint in (0, 2):
        i = [0]
        def fn():
            x = xs[i[0] % 8]; i[0] += 1
            g = ops.gemm_desc(x, w, y, M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N, residual=res, ldr=N)
            g.hint = hint
            check(lib.bist_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream), "bist_gemm")
        out[hint] = timeit(fn)
        kind = lib.bist_gemm_is_fast(C.byref(ops.gemm_desc(xs[0], w, y, M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N, residual=res, ldr=N)))
    print(f"M={M} N={N} K={K}: automatic (kind {kind}) {out[0]:.1f} us, 256-tile forced {out[2]:.1f} us")
