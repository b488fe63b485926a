"""In-kernel stamps of the 256-tile kernel (variant 4): where a workgroup's time goes (development aid)."""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops, _lib
dt = torch.bfloat16
for (M, N, K) in [(25088, 512, 512), (25088, 512, 2048), (4096, 4096, 4096)]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt); b = (torch.rand(N, K, device="cuda") * 2 - 1).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    ws = torch.zeros(1 << 20, device="cuda", dtype=torch.float32)
    kw = dict(M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=K, b_ks=1, ldc=N)
    g = ops.gemm_desc(a, b, c, **kw); g.hint = 2 + 16 * 1; g.workspace = ws.data_ptr(); g.workspace_bytes = ws.numel() * 4
    for _ in range(3):
        _lib.check(_lib.lib.bist_gemm(C.byref(g), ops._stream()), "gemm")
    torch.cuda.synchronize()
    nt = ((M + 191) // 192) * ((N + 255) // 256)
    st = ws.view(torch.int64)[: nt * 8].cpu().numpy().reshape(nt, 8).astype(np.float64) / 100.0    # us (100 MHz)
    st = st[st[:, 0] > 0]
    nt = len(st)
    t0 = st[:, 0].min()
    print(f"M={M} N={N} K={K} tiles={nt}: span {st[:,3].max()-t0:.1f} us; start spread {st[:,0].max()-t0:.1f};"
          f" prologue {np.mean(st[:,1]-st[:,0]):.2f} (max {np.max(st[:,1]-st[:,0]):.2f});"
          f" loop {np.mean(st[:,2]-st[:,1]):.2f} (min {np.min(st[:,2]-st[:,1]):.2f} max {np.max(st[:,2]-st[:,1]):.2f});"
          f" epilogue {np.mean(st[:,3]-st[:,2]):.2f} (max {np.max(st[:,3]-st[:,2]):.2f}) = setup {np.mean(st[:,4]-st[:,2]):.2f} + rows {np.mean(st[:,5]-st[:,4]):.2f} + drain {np.mean(st[:,3]-st[:,5]):.2f}; loop clock {np.mean(st[:,6]*100.0/(st[:,2]-st[:,1]))/1e3:.2f} GHz", flush=True)
