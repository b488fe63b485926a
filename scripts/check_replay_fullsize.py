"""Eager step against the replayed (hipGraph, three streams) step at BASELINE configs[1] size: parameters after ONE optimiser step from the
same weights and batch, dropout off.  A lost cross-stream dependency in the capture shows up as a full-size Adam step of difference in some
parameter; fp32-atomics order only gives differences far below the learning rate.  Development aid."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
c = bench.CFG
T = int(sys.argv[1]) if len(sys.argv) > 1 else c["T"]
args = bench.model_args(c["L"], c["d"], c["h"], 0.0)
torch.manual_seed(1)
m0 = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda()
b = synthetic_batch(c["B"], T=T, S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
if dt == torch.float32:
    b = synthetic_batch(c["B"], T=T, S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.float32)
runs = []
for g in (False, False, True, True):
    m = copy.deepcopy(m0); m.train()
    t = Trainer(m, args, c["V"], compute_dtype=dt, use_graph=g)
    t.step(b); torch.cuda.synchronize()
    runs.append((("graph" if g else "eager"), t.master.clone(), t.rate()))
    del t, m
lr = runs[0][2]
print(f"T={T} {dt}: parameters after ONE step, learning rate {lr:.3e}; fraction of elements further apart than half a learning rate:")
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        d = (runs[i][1] - runs[j][1]).abs()
        print(f"  {runs[i][0]} {i} vs {runs[j][0]} {j}: {100.0 * (d > 0.5 * lr).float().mean().item():8.4f} %   max {d.max().item():.3e}")
