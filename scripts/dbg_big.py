import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops, _lib
dt = torch.bfloat16
M = N = K = 256
a = torch.eye(M, K, device="cuda").to(dt)
for name, b in (("col", torch.arange(N, device="cuda").float()[:, None].expand(N, K).contiguous()),
                ("row", torch.arange(K, device="cuda").float()[None, :].expand(N, K).contiguous())):
    c = torch.full((M, N), -1.0, device="cuda", dtype=dt)
    g = ops.gemm_desc(a, b.to(dt).contiguous(), c, M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N); g.hint = 2
    _lib.check(_lib.lib.bist_gemm(C.byref(g), ops._stream()), "gemm")
    torch.cuda.synchronize()
    print(name)
    for m in (0, 1, 2, 17, 130):
        print(m, c[m, :72].float().int().tolist())
