"""Median time of a beam-search turn (BASELINE configs[4]) over N turns; run once per setting of the decoder switches (development aid).
usage: python scripts/bench_decode.py [turns] [history length]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
import bist_amd.model.decode as D
from bist_amd.data.synthetic import synthetic_batch

c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=int(sys.argv[2]) if len(sys.argv) > 2 else c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=torch.bfloat16)
ts = []
with torch.no_grad():
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print({k: os.environ.get(k) for k in ("BIST_DECSTACK_COLSPLIT", "BIST_DEVICE_BEAM")}, f"median {statistics.median(ts[4:]):.2f} ms, min {min(ts[4:]):.2f} ms")
