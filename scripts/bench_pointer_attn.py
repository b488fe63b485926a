"""Time of the pointer-attention launches alone (50 launches replayed from a graph), development aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import functional as Fn
from bist_amd._lib import check, lib
from bist_amd import ops
B, Lt, L, d = 16, 20, 25, 512
g = torch.Generator().manual_seed(0)
q = torch.randn(B, Lt, d, generator=g).bfloat16().cuda()
k = torch.randn(B, L, d, generator=g).bfloat16().cuda()
enc = torch.randn(B, L, d, generator=g).bfloat16().cuda()
mask = torch.ones(B, 1, L, dtype=torch.bool).cuda()
text = torch.randint(1, 50, (B, L), generator=g).cuda()
def fwd():
    with torch.no_grad():
        return Fn.pointer_attn(q, k, enc, mask, text, 0)
p, tv = fwd()
gp, gt = torch.randn_like(p), torch.randn_like(tv)
dq, dk, de = torch.empty_like(q), torch.empty_like(k), torch.empty_like(enc)
def bwd():
    check(lib.bist_pointer_attn_bwd(q.data_ptr(), k.data_ptr(), enc.data_ptr(), p.data_ptr(), gp.data_ptr(), gt.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                    de.data_ptr(), B, Lt, L, d, d ** -0.5, ops.dtype_code(q.dtype), ops._stream()), "bwd")
for name, fn in (("forward", fwd), ("backward", bwd)):
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(gr):
            for _ in range(50): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(name, "50 launches replayed:", round(e0.elapsed_time(e1) / 50 * 1e3, 2), "us per launch")
def fwd_no_tv():
    check(lib.bist_pointer_attn_fwd(q.data_ptr(), k.data_ptr(), None, None, 0, None, 0, p.data_ptr(), None, B, Lt, L, d, d ** -0.5, ops.dtype_code(q.dtype), ops._stream()), "fwd")
s = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    fwd_no_tv()
    with torch.cuda.graph(gr):
        for _ in range(50): fwd_no_tv()
torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
print("forward without the text vector (scores + softmax only):", round(e0.elapsed_time(e1) / 50 * 1e3, 2), "us per launch")
