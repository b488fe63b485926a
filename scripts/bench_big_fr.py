"""256-column-tile GEMM with FR = 5..8 row fragments (BIST_GEMM_BIG_FR) on the frame-grid shapes (development aid)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops, _lib
dt = torch.bfloat16
for (M, N, K) in [(100352, 512, 2048), (25088, 512, 2048), (25088 * 4, 512, 512)]:
    nset = max(1, int(600e6 // (2 * (M * K + M * N))))
    A = [(torch.rand(M, K, device="cuda") * 2 - 1).to(dt) for _ in range(nset)]
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nset)]
    kw = dict(M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N)
    def run(i):
        g = ops.gemm_desc(A[i % nset], b, Cs[i % nset], **kw); g.hint = 2
        _lib.check(_lib.lib.bist_gemm(C.byref(g), ops._stream()), "gemm")
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"FR={os.environ.get('BIST_GEMM_BIG_FR', 'auto')} M={M} N={N} K={K}: {us:.1f} us {2.0 * M * N * K / us / 1e6:.0f} TF", flush=True)
