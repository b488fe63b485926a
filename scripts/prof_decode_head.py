"""Head of a beam-search turn: host work before the first-step graph, its launch + execution, the rest (development aid)."""
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
sync = torch.cuda.synchronize


def med(f, n=15):
    ts = []
    for _ in range(n):
        sync(); t = time.perf_counter(); f(); sync(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[n // 2]


with torch.no_grad():
    turn = lambda: D.beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
    for _ in range(3):
        turn()
    store = D._graph_store(model)
    first = [v for k, v in store.items() if isinstance(k, tuple) and k and k[0] == "first"][0]
    print(f"whole turn                          {med(turn):.2f} ms")
    print(f"_graph_store (parameter versions)   {med(lambda: D._graph_store(model)):.2f} ms")
    print(f"first-step graph replay + sync      {med(first[0].replay):.2f} ms")
    print(f"_graph_first_step + sync            {med(lambda: D._graph_first_step(model, b1, 2, args, host=False, pad_symbol=1)):.2f} ms")
    steps = sorted((k for k in store if isinstance(k, tuple) and k and k[0] == "incr"), key=lambda k: k[1])
    print(f"one later step graph replay + sync  {med(store[steps[3]][0].replay):.2f} ms   ({len(steps)} step graphs)")

    def all_steps():
        for k in steps:
            store[k][0].replay()
    print(f"all later step graphs back to back  {med(all_steps):.2f} ms")
