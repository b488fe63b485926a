"""Beam search with the decoder layers as one persistent launch per step vs one launch per operation (development aid): n-best
lists, scores and the time per turn."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.model.decode import beam_search_decode
from bist_amd.data.synthetic import synthetic_batch

import bist_amd.model.decode as _D
_D.STEP_GRAPHS = os.environ.get("STEP_GRAPHS", "1") != "0"
c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
res = {}
for seed in (99, 100, 101):
    b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=seed, dtype=torch.bfloat16)
    for fused in (False, True):
        Fn.FUSED_DECODE = fused
        model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
        with torch.no_grad():
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res[(seed, fused)] = out
        print(f"seed {seed} fused={fused}: turn {min(ts):.2f} ms; best {out[0][0][0][:6]}... score {out[0][0][1]:.4f}")
    a, b = res[(seed, False)][0], res[(seed, True)][0]
    same = [x[0] == y[0] for x, y in zip(a, b)]
    print("  n-best token lists identical:", same, " max score diff:", max(abs(x[1] - y[1]) for x, y in zip(a, b)))
