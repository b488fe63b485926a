"""Where a beam-search turn spends its time: first-step graph, per-length step graphs, host beam logic (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
import bist_amd.model.decode as D
from bist_amd.data.synthetic import synthetic_batch

c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=torch.bfloat16)
acc = {"first": 0.0, "step": 0.0, "rows": 0.0, "n_step": 0}
f0, s0, r0, i0 = D._graph_first_step, D._graph_step, D._turn_for_rows, D._graph_step_incr
def wrap(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t
        if name == "step": acc["n_step"] += 1
        return r
    return w
D._graph_first_step, D._graph_step, D._turn_for_rows = wrap("first", f0), wrap("step", s0), wrap("rows", r0)
D._graph_step_incr = wrap("step", i0)
with torch.no_grad():
    for it in range(4):
        for k in ("first", "step", "rows"): acc[k] = 0.0
        acc["n_step"] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
        torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
        print(f"turn {it}: {tot:.2f} ms = first step {acc['first']*1e3:.2f} + {acc['n_step']} step graphs {acc['step']*1e3:.2f} + row expansion {acc['rows']*1e3:.2f} + host {tot - (acc['first']+acc['step']+acc['rows'])*1e3:.2f}")

st = model.mutlimodal_decoder.__dict__.get("_bist_dec_state")
if st is not None:
    print("decoder-stack sync words:", st["sync"].tolist(), "(word 5: 1 = write-through hand-offs, 2 = all workgroups on one XCD: plain stores)")
