"""Soak test of the persistent decoder kernel: many beam-search turns over a few dialogues, every turn's n-best compared with the
first result for that dialogue (any hand-off race in the kernel shows up as a differing list), sticky error word checked."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.model.decode import beam_search_decode
from bist_amd.data.synthetic import synthetic_batch

c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
turns = int(os.environ.get("TURNS", "300"))
dialogues = [synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=lq, Lh=lh, Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=s, dtype=torch.bfloat16)
             for s, (lq, lh) in zip((99, 100, 101, 102, 103, 104), ((20, 60), (20, 60), (12, 45), (20, 60), (17, 130), (9, 250)))]     # (the last two: the kernel's chunked core)
first, bad = {}, 0
t0 = time.time()
with torch.no_grad():
    for it in range(turns):
        k = it % len(dialogues)
        out = beam_search_decode(model, dialogues[k], 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)[0]
        sig = [(tuple(int(t) for t in h[0]), round(float(h[1]), 4)) for h in out]
        if k not in first:
            first[k] = sig
        elif sig != first[k]:
            bad += 1
            print(f"turn {it}: dialogue {k} differs from its first result", flush=True)
        if it % 50 == 49:
            print(f"{it + 1} turns, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
st = model.mutlimodal_decoder.__dict__["_bist_dec_state"]
print("mismatches:", bad, " sync words:", st["sync"].tolist())
