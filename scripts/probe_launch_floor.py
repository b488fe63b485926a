import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
x = torch.zeros(64, device="cuda"); y = torch.zeros(64, device="cuda", dtype=torch.bfloat16)
def chain(n):
    for _ in range(n): ops.cast(x, torch.bfloat16)
for mode in ("dependent",):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): chain(3)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): chain(200)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("graph replay, 200 trivial kernels: %.2f us per kernel" % (e0.elapsed_time(e1) / 200 * 1e3))
e0.record(); chain(200); e1.record(); torch.cuda.synchronize()
print("eager, 200 trivial kernels: %.2f us per kernel" % (e0.elapsed_time(e1) / 200 * 1e3))
