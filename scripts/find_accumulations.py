"""Autograd nodes of one training step whose output gradient is accumulated from several consumers (each costs an aten add launch and an
accumulation point on a backward chain): walk the graph from the loss and count incoming edges per (node, output).  Development aid."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda(); model.train()
tr = Trainer(model, args, 3000, compute_dtype=torch.bfloat16, use_graph=False)
b = synthetic_batch(16, dtype=torch.bfloat16, seed=1)
loss, terms = tr.forward_loss(b)
indeg = collections.Counter()
consumers = collections.defaultdict(list)
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or id(n) in seen:
        continue
    seen.add(id(n))
    for nxt, idx in n.next_functions:
        if nxt is None:
            continue
        indeg[(nxt, idx)] += 1
        consumers[(id(nxt), idx)].append(n.name())
        stack.append(nxt)
rows = [(k, v) for k, v in indeg.items() if v > 1 and "AccumulateGrad" not in k[0].name()]
print(len(seen), "nodes;", len(rows), "accumulated outputs")
for (n, idx), v in sorted(rows, key=lambda kv: kv[0][0].name()):
    meta = ""
    try:
        meta = str([tuple(m.shape) for m in n._input_metadata][:3])
    except Exception:
        pass
    print(f"{n.name():34s} out {idx}  consumers {v}: {sorted(consumers[(id(n), idx)])}  {meta}")
