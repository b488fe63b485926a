"""What does a resident wave on another hardware queue cost a chain of small dependent launches?  (development aid, round 4)"""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS, ops
from bist_amd._lib import lib, check
s0, s1, s2, s3 = GS.distinct_streams(4)
w = torch.zeros(8, dtype=torch.int64, device="cuda")
x = torch.randn(320, 512, device="cuda", dtype=torch.bfloat16)
a = torch.ones(512, device="cuda", dtype=torch.bfloat16); bb = torch.zeros(512, device="cuda", dtype=torch.bfloat16)
wt = torch.randn(512, 512, device="cuda", dtype=torch.bfloat16) * 0.05
big = torch.randn(25088, 512, device="cuda", dtype=torch.bfloat16)


def chain(kind, n):
    y = x
    for _ in range(n):
        if kind == "ln":
            y = ops.layernorm(y, a, bb)
        elif kind == "gemm":
            y = ops.linear(y, wt, None)
        elif kind == "biggemm":
            ops.linear(big, wt, None)
    return y


def timed(kind, n, resident, graph):
    with torch.cuda.stream(s0):
        chain(kind, 5)
        g = None
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s0, capture_error_mode="thread_local"):
                chain(kind, n)
            g.replay()
        torch.cuda.synchronize()
        for st, mode in resident:
            check(lib.bist_dev_idle_wave(st.cuda_stream, int(30e-3 * 1e8), mode, w.data_ptr()), "idle")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if g is not None:
            g.replay()
        else:
            chain(kind, n)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n


for kind, n in (("ln", 400), ("gemm", 400), ("biggemm", 100)):
    for graph in (True,):
        base = timed(kind, n, [], graph)
        one = timed(kind, n, [(s1, 3)], graph)
        two = timed(kind, n, [(s1, 3), (s2, 3)], graph)
        three = timed(kind, n, [(s1, 3), (s2, 3), (s3, 3)], graph)
        print(f"{kind:8s} graph={graph}: {base:7.2f} us per launch alone; {one:7.2f} with one resident wave; {two:7.2f} with two; {three:7.2f} with three", flush=True)
