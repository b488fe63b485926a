"""Is the replayed training step bound by the host's hipGraphLaunch or by the device?  (development aid, round 4)

  a. 20 steps back to back (the bench's loop)
  b. every step issued BEHIND a long spin kernel on the launch stream: the host has enqueued the whole graph before the device may start,
     so the device time of the step (HIP events) is what the device needs when it never waits for the host
  c. two instantiations of the step replayed in alternation (a launch of exec B does not have to wait for exec A's previous run)
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer

c = dict(bench.CFG)
c["T"] = int(os.environ.get("T", c["T"]))
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
for _ in range(5): tr.step(b)
torch.cuda.synchronize()


def loop(n, step):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n


for rep in range(3):
    issue, wall = loop(20, lambda i: tr.step(b))
    print(f"a. back to back: host issue {issue:.2f} ms/step, wall {wall:.2f} ms/step", flush=True)

# b. behind a spin kernel
cyc = int(2.0e9 * 0.030)      # ~30 ms at ~2 GHz
dev = []
for rep in range(8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    torch.cuda._sleep(cyc)
    e0.record()
    tr.step(b)
    e1.record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    dev.append((e0.elapsed_time(e1), (h1 - h0) * 1e3))
print("b. behind a 30 ms spin kernel: (device ms of the step, host ms to issue) " + " ".join(f"({d:.2f},{h:.2f})" for d, h in dev), flush=True)

# c. two instantiations in alternation
sets = [(tr._graph, tr._graph2, tr._graph_key, tr._static_batch, tr._static_terms, tr._static_src)]
tr._graph = None
tr.step(b)              # captures a second set
sets.append((tr._graph, tr._graph2, tr._graph_key, tr._static_batch, tr._static_terms, tr._static_src))
torch.cuda.synchronize()


def pingpong(i):
    tr._graph, tr._graph2, tr._graph_key, tr._static_batch, tr._static_terms, tr._static_src = sets[i & 1]
    tr.step(b)


for _ in range(4): pingpong(_)
for rep in range(3):
    issue, wall = loop(20, pingpong)
    print(f"c. two execs in alternation: host issue {issue:.2f} ms/step, wall {wall:.2f} ms/step", flush=True)
tr._graph, tr._graph2, tr._graph_key, tr._static_batch, tr._static_terms, tr._static_src = sets[0]
for rep in range(2):
    issue, wall = loop(20, lambda i: tr.step(b))
    print(f"a'. back to back again: host issue {issue:.2f} ms/step, wall {wall:.2f} ms/step", flush=True)
