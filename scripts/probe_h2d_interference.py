"""What does a 205 MB host-to-device copy per step cost the replayed training step when nothing waits for it? (development aid, round 4:
the PCIe-inclusive `fed` step is 2.7 ms longer than the resident one although the copy alone takes 3.6 ms of an 8 ms step)."""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn, ops
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
c = dict(bench.CFG)
if Fn.main_stream() is not None:
    torch.cuda.set_stream(Fn.main_stream())
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
n = c["B"] * c["T"] * c["S"] * c["C"]
host = torch.empty(n, dtype=torch.float32).pin_memory(); host.normal_()
dev = torch.empty(n, dtype=torch.float32, device="cuda")
dev16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
for _ in range(5): tr.step(b)
torch.cuda.synchronize()
copy = Fn.copy_stream() or torch.cuda.Stream()
pool = torch.cuda.Stream()


def run(label, per_step):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        per_step()
        tr.step(b)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) * 50:.3f} ms per step", flush=True)


def h2d(stream, pieces=1, cast=False):
    def f():
        with torch.cuda.stream(stream):
            k = n // pieces
            for p in range(pieces):
                dev[p * k:(p + 1) * k].copy_(host[p * k:(p + 1) * k], non_blocking=True)
            if cast:
                dev16.copy_(dev)
    return f


run("no copy", lambda: None)
run("205 MB H2D per step on the feeder's copy stream", h2d(copy))
run("... in 8 pieces", h2d(copy, 8))
run("... on a pool stream", h2d(pool))
run("... + cast to bf16 on the copy stream", h2d(copy, 1, True))
run("no copy", lambda: None)
