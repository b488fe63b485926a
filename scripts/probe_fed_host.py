"""Host time of the PCIe-inclusive loop per step: the feeder's next() and the trainer's step() (development aid, round 4)."""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.feeder import DeviceFeeder, HostBatch
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
c = dict(bench.CFG)
if Fn.main_stream() is not None:
    torch.cuda.set_stream(Fn.main_stream())
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
host = []
for i in range(2):
    hb = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234 + i, dtype=torch.float32, device="cpu")
    pf = DeviceFeeder.pinned_like(hb.fts.shape, hb.fts.dtype); pf.copy_(hb.fts)
    host.append(HostBatch(hb.query, hb.his, pf, hb.cap, hb.trg, hb.trg_y))
n = 24
it = iter(DeviceFeeder([host[i % 2] for i in range(n + 4)], feature_dtype=torch.bfloat16))
for _ in range(4):
    tr.step(next(it))
torch.cuda.synchronize()
tn, ts = [], []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); fb = next(it); b_ = time.perf_counter(); tr.step(fb); c_ = time.perf_counter()
    tn.append((b_ - a) * 1e3); ts.append((c_ - b_) * 1e3)
torch.cuda.synchronize()
print(f"fed loop: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step; host in next(): median {sorted(tn)[n // 2]:.3f} ms (max {max(tn):.3f}); host in step(): median {sorted(ts)[n // 2]:.3f} ms (max {max(ts):.3f})")
print("next() ms:", [round(x, 2) for x in tn])
fd = DeviceFeeder([host[i % 2] for i in range(14)], feature_dtype=torch.bfloat16)
it = iter(fd)
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
main = torch.cuda.current_stream()
base = E(); base.record(main)
for i in range(13):
    a0 = E(); a0.record(fd.copy_stream)           # the copy stream reaches the point where this next() queues its work
    fb = next(it)
    a1 = E(); a1.record(fd.copy_stream)           # ... and finishes it (the staged batch is the one AFTER the batch returned)
    s0 = E(); s0.record(main)
    tr.step(fb)
    s1 = E(); s1.record(main)
    rows.append((a0, a1, s0, s1))
torch.cuda.synchronize()
print("per iteration, ms since the first: copy stream [begin, end] of the work queued by next(); main stream [begin, end] of step()")
for i, (a0, a1, s0, s1) in enumerate(rows):
    print("  %2d  copy %8.2f .. %8.2f   step %8.2f .. %8.2f" % (i, base.elapsed_time(a0), base.elapsed_time(a1), base.elapsed_time(s0), base.elapsed_time(s1)))
it = iter(DeviceFeeder([host[i % 2] for i in range(24)], feature_dtype=torch.bfloat16))
torch.cuda.synchronize(); t0 = time.perf_counter(); k = 0
for fb in it:
    fb.fts.sum()          # a consumer that takes no time
    k += 1
torch.cuda.synchronize()
print(f"feeder alone: {(time.perf_counter() - t0) / k * 1e3:.3f} ms per batch")
import cProfile, pstats
it = iter(DeviceFeeder([host[i % 2] for i in range(12)], feature_dtype=torch.bfloat16))
for _ in range(4):
    tr.step(next(it))
pr = cProfile.Profile()
for _ in range(6):
    pr.enable(); fb = next(it); pr.disable()
    tr.step(fb)
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
