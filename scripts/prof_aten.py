"""Which aten ops (torch-launched kernels) remain in one training step (development aid)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda(); model.train()
tr = Trainer(model, args, 3000, compute_dtype=torch.bfloat16, use_graph=False)
b = synthetic_batch(16, dtype=torch.bfloat16, seed=1)
for _ in range(2): tr.backward(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.backward(b)
torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
skip = ("aten::view", "aten::reshape", "aten::slice", "aten::as_strided", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::view_as", "aten::transpose", "aten::permute", "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::expand",
        "aten::detach", "aten::alias", "aten::t", "aten::_unsafe_view", "aten::unflatten", "aten::stride", "aten::size", "aten::is_nonzero",
        "aten::resize_", "aten::set_", "aten::result_type", "aten::to", "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense")
for e in [r for r in rows if r.key not in skip][:60]:
    print(f"{e.key:32s} n={e.count:4d}  shapes={str(e.input_shapes)[:110]}")
