"""Interference matrix between hardware queues: a single-queue graph of 300 dependent small launches on stream i, timed alone and beside a
resident wave (with K launches queued behind it) on stream j.  (development aid, round 4)"""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS, ops
from bist_amd._lib import lib, check
N = int(os.environ.get("NSTREAMS", "8"))
K = int(os.environ.get("PENDING", "1"))
streams = GS.distinct_streams(N)
x = torch.randn(320, 512, device="cuda", dtype=torch.bfloat16)
a = torch.ones(512, device="cuda", dtype=torch.bfloat16); bb = torch.zeros(512, device="cuda", dtype=torch.bfloat16)
w = torch.zeros(8, dtype=torch.int64, device="cuda")
graphs = []
for s in streams:
    with torch.cuda.stream(s):
        y = x
        for _ in range(3): y = ops.layernorm(y, a, bb)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            y = x
            for _ in range(300): y = ops.layernorm(y, a, bb)
        g.replay()
    graphs.append(g)
torch.cuda.synchronize()


def run(i, j):
    torch.cuda.synchronize()
    if j is not None:
        check(lib.bist_dev_idle_wave(streams[j].cuda_stream, int(4e-3 * 1e8), 0, w.data_ptr()), "idle")
        for _ in range(K):
            check(lib.bist_dev_idle_wave(streams[j].cuda_stream, 10, 0, w.data_ptr()), "idle")
    with torch.cuda.stream(streams[i]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graphs[i].replay(); e1.record()
    streams[i].synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 300
    torch.cuda.synchronize()
    return t


print(f"us per dependent launch of a 300-launch single-queue graph on stream i (rows) beside a resident wave + {K} queued launches on stream j (columns); first column: alone")
for i in range(N):
    row = [run(i, None)] + [run(i, j) if j != i else float("nan") for j in range(N)]
    print(f"{i}: " + " ".join(f"{v:5.2f}" for v in row), flush=True)
