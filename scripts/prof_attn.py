"""The roofline region of bench.py (P0 + LayerNorm + one VidEncoderLayer4, inference) run eagerly for rocprofv3.  Two warm-up
passes, then `--iters` passes bracketed by two spin kernels (torch.cuda._sleep): scripts/pmc_region.py and
scripts/region_kernels.py cut the traces at those markers."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--T", type=int, default=32)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
args = argparse.Namespace(d_model=512, att_h=8, nb_blocks=1, nb_venc_blocks=1, nb_cenc_blocks=1, nb_aenc_blocks=0,
                          t2s=1, s2t=1, ptr_gen=1, ptr_ft="query,cap", mask_unk=1, auto_encoder=1, include_caption="summary",
                          enc_st_combine="none", dec_st_combine="seq", enc_vc_combine="dyn", dropout=0.0, d_ff=2048)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda().to(torch.bfloat16).eval()
b = synthetic_batch(a.B, T=a.T, dtype=torch.bfloat16)
with torch.no_grad():
    q = model.encode_text(b, {})["encoded_query"]
    vl = model.mutlimodal_decoder.v_layers[0]

    def run():
        f = model.vid_encoder(b, {})
        vl({"t2s": q, "s2t": q}, f, b)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
print("done")
