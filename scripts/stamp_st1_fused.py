"""Phase timeline of st1_fused_kernel from in-kernel s_memtime stamps (development aid; bist_dev_set_stamps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(os.environ.get("B", "64")); T = int(os.environ.get("T", "32")); S = int(os.environ.get("S", "49")); Lq, d, h = 20, 512, 8
direction = int(os.environ.get("DIR", "0"))
G = S if direction == 0 else T
K = T if direction == 0 else S
NG = 4 if K <= 32 else 2 if K <= 64 else 1
nwg = B * ((G + NG - 1) // NG)
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
from bist_amd import _lib as _bl
_bl.check(_bl.lib.bist_dev_set_stamps(0, stamps.data_ptr()), "bist_dev_set_stamps")      # explicit hand-over of a buffer this script owns
from bist_amd import ops

g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda()
vft, qf, x = r(B, T, S, d), r(B, Lq * h, d, sc=d ** -0.5), r(B, Lq, d)
wv, bv, wo, bo = ops.pack_frag_rows(r(d, d, sc=d ** -0.5)), r(d, sc=0.1), ops.pack_frag_rows(r(d, d, sc=d ** -0.5)), r(d, sc=0.1)
out = torch.empty(B, G, Lq, d, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    ops.st_stage1_fused(qf, vft, None, wv, bv, wo, bo, x, h=h, direction=direction, out=out)
torch.cuda.synchronize()
st = stamps.view(nwg, 8).cpu().double()
t0 = st[:, 0].min()
rel = (st - t0) / 100.0          # s_memtime ticks at 100 MHz -> us
names = ["start", "X landed", "V done", "S done", "barrier", "ctx done", "out-proj", "end"]
dur = rel[:, 1:] - rel[:, :-1]
print(f"dir={direction} B={B} T={T} S={S}: {nwg} workgroups; kernel span {float(rel[:, 7].max()):.1f} us")
print("phase durations (us): median / p90 over workgroups")
for i in range(7):
    print(f"  {names[i]:>9s} -> {names[i + 1]:<9s} {float(dur[:, i].median()):7.2f} {float(dur[:, i].quantile(0.9)):7.2f}")
print(f"  workgroup total       {float((rel[:, 7] - rel[:, 0]).median()):7.2f} {float((rel[:, 7] - rel[:, 0]).quantile(0.9)):7.2f}")
starts = rel[:, 0].sort().values
print("start times (us) of workgroups #0, #255, #256, #511, #512, last:", [round(float(starts[min(i, nwg - 1)]), 1) for i in (0, 255, 256, 511, 512, nwg - 1)])
