"""Cost of NEW dialogue geometries in beam_search_decode: turns over dialogues of different (query, history, caption) lengths -- first
sight of a geometry (graph captures) against its replays (development aid).  usage: python scripts/decode_geometries.py"""
import os, sys, time
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
geoms = [(20, 60, 25), (17, 43, 25), (12, 80, 19), (20, 60, 25), (17, 43, 25), (23, 31, 22), (12, 80, 19), (23, 31, 22)]
with torch.no_grad():
    for Lq, Lh, Lc in geoms:
        b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=Lq, Lh=Lh, Lc=Lc, Lt=c["Lt"], vocab=c["V"], seed=Lq * 1000 + Lh, dtype=torch.bfloat16)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
        torch.cuda.synchronize()
        print(f"(Lq, Lh, Lc) = {(Lq, Lh, Lc)}: {(time.perf_counter() - t0) * 1e3:8.1f} ms, graphs held {len(model.__dict__.get('_bist_step_graphs', {}))}")
