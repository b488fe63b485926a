"""Host time of one replay of the training step's hipGraph against its device time (development aid): if hipGraphLaunch takes longer than
the step runs, the step is bound by the host."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
c = bench.CFG
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
for _ in range(5): tr.step(b)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    h0 = time.perf_counter(); tr.step(b); host.append(time.perf_counter() - h0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"20 steps: host time to issue {t_issue * 1e3 / 20:.2f} ms per step (median call {sorted(host)[10] * 1e3:.2f} ms), wall {t_all * 1e3 / 20:.2f} ms per step")

# the same with the device idle at every call: the launch's own host time, and the time until the step has finished on the device
call, done = [], []
for _ in range(12):
    torch.cuda.synchronize()
    h0 = time.perf_counter(); tr.step(b); h1 = time.perf_counter(); torch.cuda.synchronize(); h2 = time.perf_counter()
    call.append(h1 - h0); done.append(h2 - h0)
print(f"idle device: call {sorted(call)[6] * 1e3:.2f} ms, finished {sorted(done)[6] * 1e3:.2f} ms after the call began")
