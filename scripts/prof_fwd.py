"""Forward-only timing of the HIP model at the benchmark geometry (development aid, not the bench)."""
import argparse
import sys
import os
import time
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bist_amd.model as M
from bist_amd import ops
from bist_amd.data.synthetic import synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--L", type=int, default=6)
ap.add_argument("--T", type=int, default=32)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--gemm-timing", type=int, default=1)
a = ap.parse_args()
dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
args = argparse.Namespace(d_model=512, att_h=8, nb_blocks=a.L, nb_venc_blocks=a.L, nb_cenc_blocks=a.L, nb_aenc_blocks=0,
                          t2s=1, s2t=1, ptr_gen=1, ptr_ft="query,cap", mask_unk=1, auto_encoder=1, include_caption="summary",
                          enc_st_combine="none", dec_st_combine="seq", enc_vc_combine="dyn", dropout=0.0, d_ff=2048)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda().to(dtype).eval()
b = synthetic_batch(a.B, T=a.T, dtype=dtype)
with torch.no_grad():
    for _ in range(2):
        ft = model.forward(b); lp = model.generator(ft, b, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ft = model.forward(b); lp = model.generator(ft, b, args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"B={a.B} T={a.T} L={a.L} {a.dtype}: forward+generator {dt*1e3:.3f} ms  ({a.B/dt:.1f} clips/s)")
    if a.gemm_timing:
        ops.GEMM_TIMING = []
        ft = model.forward(b); lp = model.generator(ft, b, args)
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for tag, e0, e1, _plan in ops.GEMM_TIMING:
            agg.setdefault(tag, []).append(e0.elapsed_time(e1))
        tot = 0
        for tag, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            M_, N_, K_, Z_ = tag
            fl = 2.0 * M_ * N_ * K_ * Z_
            ms = sum(v) / len(v)
            tot += sum(v)
            print(f"  gemm M={M_:7d} N={N_:5d} K={K_:5d} z={Z_:3d} x{len(v):3d}: {ms*1e3:9.1f} us avg  {fl/ms/1e9:8.1f} TFLOP/s   total {sum(v):.3f} ms")
        print(f"  all GEMM launches: {tot:.3f} ms (event-bracketed; includes launch gaps)")
