"""A test-set-like sweep of beam_search_decode: dialogues of ten turns each, the history growing turn by turn, query / caption lengths
drawn per dialogue (generate.py:30-60 iterates such a set) -- total time, average per turn, graph captures and device memory held, for
the length-bucket and graph-store settings in the environment (development aid).
usage: [BIST_DECODE_BUCKET=8|16|class] [BIST_DECODE_MAX_GEOMETRIES=n] python scripts/decode_eval_sweep.py [dialogues]"""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
import bist_amd.model.decode as D
from bist_amd.data.synthetic import synthetic_batch

n_dialogues = int(sys.argv[1]) if len(sys.argv) > 1 else 30
c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
rng = random.Random(7)
turns = []
for dlg in range(n_dialogues):
    Lc = rng.randint(12, 40)
    Lh = 0
    for t in range(10):
        Lq = rng.randint(6, 22)
        turns.append((Lq, max(Lh, 2), Lc))               # (turn 0: the reference feeds a 2-token empty history)
        Lh = min(Lh + Lq + rng.randint(4, 14), 250)      # question + answer appended to the history
base = torch.cuda.memory_allocated()
seen, times, mismatches = set(), [], 0
with torch.no_grad():
    t_all = time.perf_counter()
    for i, (Lq, Lh, Lc) in enumerate(turns):
        b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=Lq, Lh=Lh, Lc=Lc, Lt=c["Lt"], vocab=c["V"], seed=i, dtype=torch.bfloat16)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)[0]
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        seen.add((Lq, Lh, Lc))
        if i % 7 == 3:                                    # (untimed) the same dialogue again, through whatever graphs the turn left: the same n-best
            again = D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)[0]
            sig = lambda hs: [(tuple(int(t) for t in h[0]), round(float(h[1]), 4)) for h in hs]
            if sig(again) != sig(out):
                mismatches += 1
                print(f"turn {i} {(Lq, Lh, Lc)}: the repeated turn differs", flush=True)
    t_all = time.perf_counter() - t_all
times_ms = sorted(t * 1e3 for t in times)
held = sum(1 for k in model.__dict__.get("_bist_step_graphs", {}) if isinstance(k, tuple) and k and k[0] == "first")
print(f"bucket {D.BUCKET}, store bound {D.MAX_GEOMETRIES}: {len(turns)} turns, {len(seen)} exact geometries: {sum(times) :.1f} s in decode, "
      f"{sum(times) * 1e3 / len(turns):.1f} ms per turn (median {times_ms[len(times_ms) // 2]:.1f}, max {times_ms[-1]:.0f}), "
      f"{sum(1 for t in times_ms if t > 25)} turns with captures, {held} first-step graphs held, "
      f"{(torch.cuda.memory_allocated() - base) / 2**20:.0f} MiB held, {torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved; "
      f"{mismatches} repeated turns differed")
model.mutlimodal_decoder.check_decode_errors()
