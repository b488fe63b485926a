"""Big K-contiguous products: 128-tile kernel vs the 256x256 kernel (hint) vs hipBLASLt (yardstick), random operands,
rotating operand sets larger than the 256 MB MALL when COLD=1 (development aid)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops, _lib
dt = torch.bfloat16
SH = [(25088, 512, 2048), (25088, 512, 512), (25088, 2048, 512), (15680, 512, 512), (4096, 4096, 4096), (100352, 512, 2048)]
if os.environ.get("SWEEP"):
    SH = [(m, 512, k) for k in (512, 2048) for m in (12544, 15680, 18816, 25088, 31360, 37632, 50176, 62720, 100352)] + [(25088, 1024, 512), (25088, 1536, 512), (15680, 2048, 512), (15680, 512, 2048)]
cold = bool(os.environ.get("COLD"))
for (M, N, K) in SH:
    nset = max(1, int(600e6 // (2 * (M * K + M * N)))) if cold else 1
    A = [(torch.rand(M, K, device="cuda") * 2 - 1).to(dt) for _ in range(nset)]
    b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nset)]
    kw = dict(M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N)
    def run(hint, i):
        g = ops.gemm_desc(A[i % nset], b, Cs[i % nset], **kw); g.hint = hint
        _lib.check(_lib.lib.bist_gemm(C.byref(g), ops._stream()), "gemm")
    def blas(i):
        torch.mm(A[i % nset], b.t(), out=Cs[i % nset])
    res = []
    ref = None
    cases = [("blas", blas), ("auto", lambda i: run(0, i)), ("t256", lambda i: run(2, i))]
    for name, fn in cases:
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        if name == "blas": ref = Cs[0].float()
        else: name += f" err {(Cs[0].float() - ref).abs().max().item():.3g}"
        it = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it): fn(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        res.append(f"{name} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")
    print(f"M={M:6d} N={N:4d} K={K:4d} sets={nset}: " + " | ".join(res), flush=True)
