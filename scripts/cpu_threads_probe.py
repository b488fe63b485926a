"""Oracle training-step time against the thread count on this host (development aid for bench.py's cpu_baseline)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bist_oracle as O
cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=6, nb_venc_blocks=6, nb_cenc_blocks=6)
torch.manual_seed(0)
sd = {k: (torch.randn(s) * 0.02).requires_grad_(True) for k, s in O.state_shapes(cfg, 3000, 2048).items()}
for a in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj"):
    sd[a] = sd["query_embed.0.lut.weight"]
ob = O.det_batch(16, 32, 49, 2048, 20, 60, 25, 20, 3000)
for n in (16, 32, 48, 64, 128):
    torch.set_num_threads(n)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        ft = O.mtn_forward(sd, cfg, ob)
        O.loss_compute(sd, cfg, ft, ob, 3000)["total"].backward()
        for v in sd.values():
            v.grad = None
        ts.append(time.perf_counter() - t0)
    print(n, [round(t, 2) for t in ts], flush=True)
