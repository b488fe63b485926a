"""bist_gemm_pair against two separate launches on the backward shapes of a small linear layer (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
from bist_amd._lib import lib, check
import ctypes as C
dt = torch.bfloat16
def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rows, n_out, k_in in [(320, 512, 512), (320, 1536, 512), (320, 2048, 512), (320, 512, 2048), (400, 1024, 512)]:
    dz = torch.randn(rows, n_out, device="cuda").to(dt); w = torch.randn(n_out, k_in, device="cuda").to(dt); x = torch.randn(rows, k_in, device="cuda").to(dt)
    dx = torch.empty(rows, k_in, device="cuda", dtype=dt); dw = torch.zeros(n_out, k_in, device="cuda", dtype=dt)
    ga = ops.gemm_desc(dz, w, dx, M=rows, N=k_in, K=n_out, a_rs=n_out, a_ks=1, b_rs=1, b_ks=k_in, ldc=k_in)
    gb = ops.gemm_desc(dz, x, dw, M=n_out, N=k_in, K=rows, a_rs=1, a_ks=n_out, b_rs=1, b_ks=k_in, ldc=k_in, residual=dw, ldr=k_in)
    s = torch.cuda.current_stream
    tp = timeit(lambda: ops.gemm_pair(ga, gb))
    def sep():
        check(lib.bist_gemm(C.byref(ga), s().cuda_stream), "a"); check(lib.bist_gemm(C.byref(gb), s().cuda_stream), "b")
    ts = timeit(sep)
    ta = timeit(lambda: check(lib.bist_gemm(C.byref(ga), s().cuda_stream), "a"))
    tb = timeit(lambda: check(lib.bist_gemm(C.byref(gb), s().cuda_stream), "b"))
    print(f"rows={rows} n_out={n_out} k_in={k_in}: pair {tp:.1f} us, separate {ts:.1f} us (dX {ta:.1f} + dW {tb:.1f})")
