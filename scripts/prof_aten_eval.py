"""Which aten ops (torch-launched kernels) remain in the inference pass over the roofline region (development aid)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
import bench
args = bench.model_args(1, 512, 8, 0.0)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda().to(torch.bfloat16).eval()
b = synthetic_batch(int(os.environ.get("B", "16")), dtype=torch.bfloat16, seed=1)
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    q = model.encode_text(b, {})["encoded_query"]
    vl = model.mutlimodal_decoder.v_layers[0]
    def run():
        f = model.vid_encoder(b, {})
        vl({"t2s": q, "s2t": q}, f, b)
    for _ in range(2): run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        run()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
skip = ("aten::view", "aten::reshape", "aten::slice", "aten::as_strided", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::view_as", "aten::transpose", "aten::permute", "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::expand",
        "aten::detach", "aten::alias", "aten::t", "aten::_unsafe_view", "aten::unflatten", "aten::stride", "aten::size", "aten::is_nonzero",
        "aten::resize_", "aten::set_", "aten::result_type", "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense")
for e in [r for r in rows if r.key not in skip][:40]:
    print(f"{e.key:32s} n={e.count:4d}  shapes={str(e.input_shapes)[:110]}")
