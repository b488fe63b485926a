import os, sys
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS
scratch = torch.zeros(4, dtype=torch.int64, device="cuda")
pool = []
for _ in range(8):
    pool.append(torch.cuda.Stream())
print("handles", [hex(s.cuda_stream) for s in pool])
for i, a in enumerate(pool):
    row = [GS._pace(a.cuda_stream, None, scratch)] + [GS._pace(a.cuda_stream, b.cuda_stream, scratch) if b is not a else float("nan") for b in pool]
    print(i, " ".join(f"{v:5.2f}" for v in row), flush=True)
got = GS.distinct_streams(4)
print("picked", [hex(s.cuda_stream) for s in got], [[h.cuda_stream for h in pool].index(s.cuda_stream) if s.cuda_stream in [h.cuda_stream for h in pool] else -1 for s in got])
