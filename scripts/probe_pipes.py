"""How many streams can the probe find on mutually independent hardware queues?  And does a long-lived wait on a 4th stream slow the step?"""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS
scratch = torch.zeros(4, dtype=torch.int64, device="cuda")
pool = []
seen = set()
for _ in range(40):
    s = torch.cuda.Stream()
    if s.cuda_stream not in seen:
        seen.add(s.cuda_stream); pool.append(s)
print(len(pool), "distinct torch streams")
n = min(len(pool), 16)
mat = [[1 if i == j else int(GS._distinct(pool[i].cuda_stream, pool[j].cuda_stream, scratch)) for j in range(n)] for i in range(n)]
for r in mat:
    print("".join(str(v) for v in r))
for want in (3, 4, 5, 6):
    try:
        GS._EXEC_STREAMS.clear()
        t0 = time.perf_counter()
        got = GS.distinct_streams(want)
        print(want, "->", len(got), f"{(time.perf_counter() - t0) * 1e3:.0f} ms")
    except Exception as e:
        print(want, "-> failed:", e)
