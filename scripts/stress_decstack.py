"""bist_decoder_stack_fwd against the layer-by-layer path for every row count of a beam-search turn, repeated (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.batch import subsequent_mask
from bist_amd.model.decode import _turn_for_rows
from bist_amd.data.synthetic import synthetic_batch

c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
b = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=torch.bfloat16)
reps = int(os.environ.get("REPS", "6"))
with torch.no_grad():
    ft = model.encode(b)
    b.trg, b.trg_mask = b.trg[:1, :1].contiguous(), subsequent_mask(1, "cuda")
    ft = model.decode(b, ft)
    for n in (1, 5):
        for Lt in range(1, 13):
            g = torch.Generator().manual_seed(n * 100 + Lt)
            trg = torch.randint(4, c["V"], (n, Lt), generator=g).cuda()
            outs = []
            for fused in [False] + [True] * reps:
                Fn.FUSED_DECODE = fused
                bn, fn = _turn_for_rows(b, ft, n, {})
                bn.trg, bn.trg_mask = trg, subsequent_mask(Lt, "cuda")
                outs.append(model.decode(bn, dict(fn))["decoded_text"].float().cpu())
            errs = [(o - outs[0]).abs().max().item() for o in outs[1:]]
            rep = max((o - outs[1]).abs().max().item() for o in outs[1:])
            print(f"n={n} Lt={Lt:2d} R={n*Lt:2d}: max err vs unfused {max(errs):.4f} (min {min(errs):.4f}); fused run-to-run {rep:.4f}", flush=True)
