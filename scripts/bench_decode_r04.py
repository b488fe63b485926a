"""Beam-search turn timing alone (development aid): python scripts/bench_decode_r04.py ; BIST_SPLIT_GRAPH=0/1."""
import os, sys, time, json
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.model import decode as D
c = dict(bench.CFG)
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
if Fn.main_stream() is not None and os.environ.get("NULLSTREAM") != "1":
    torch.cuda.set_stream(Fn.main_stream())
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
for Lh in (60, 200):
    b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=Lh, Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=torch.bfloat16)
    with torch.no_grad():
        for _ in range(2):
            res0 = D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
        before = dict(D.STATS)
        ts = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    srt = sorted(ts)
    print(f"history {Lh}: median {srt[10]:.3f} ms, p90 {srt[18]:.3f}, min {srt[0]:.3f}, max {srt[-1]:.3f}; did {({k: D.STATS[k] - before[k] for k in before})}; "
          f"same n-best as the capture turn: {[r[0] for r in res[0]] == [r[0] for r in res0[0]]}", flush=True)
