"""Do launch-bound chains on independent hardware queues slow each other down?  k single-queue graphs of dependent small launches replayed at
the same time on k independent streams; us per launch of each.  (development aid, round 4)"""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS, ops
streams = GS.distinct_streams(4)
x = torch.randn(320, 512, device="cuda", dtype=torch.bfloat16)
a = torch.ones(512, device="cuda", dtype=torch.bfloat16); bb = torch.zeros(512, device="cuda", dtype=torch.bfloat16)
wt = torch.randn(512, 512, device="cuda", dtype=torch.bfloat16) * 0.05
N = 400


def build(kind):
    gs = []
    for s in streams:
        with torch.cuda.stream(s):
            y = x
            for _ in range(3):
                y = ops.layernorm(y, a, bb) if kind == "ln" else ops.linear(y, wt, None)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                y = x
                for _ in range(N):
                    y = ops.layernorm(y, a, bb) if kind == "ln" else ops.linear(y, wt, None)
            g.replay()
        gs.append(g)
    torch.cuda.synchronize()
    return gs


for kind in ("ln", "gemm"):
    gs = build(kind)
    for k in (1, 2, 3, 4):
        torch.cuda.synchronize()
        ev = []
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gs[i].replay(); e1.record()
                ev.append((e0, e1))
        torch.cuda.synchronize()
        print(f"{kind}: {k} chains at once: us per launch " + " ".join(f"{e0.elapsed_time(e1) * 1e3 / N:5.2f}" for e0, e1 in ev), flush=True)
