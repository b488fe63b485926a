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


from bist_amd._lib import lib, check
from bist_amd.ops import dtype_code


def cast_only(stream):
    def f():
        check(lib.bist_cast(dev.data_ptr(), dev16.data_ptr(), n, dtype_code(torch.float32), dtype_code(torch.bfloat16), stream.cuda_stream), "bist_cast")
    return f


def h2d_cast(stream):
    g, cst = h2d(stream), cast_only(stream)
    def f():
        g(); cst()
    return f


gen = [0]
def new_fts():
    gen[0] += 1
    b.fts._bist_generation = gen[0]          # the trainer copies the field into the graph's static buffer again (as for a fed batch)


def h2d_event_wait(stream, with_cast):
    g = h2d_cast(stream) if with_cast else h2d(stream)
    def f():
        g()
        ev = torch.cuda.Event(); ev.record(stream)
        torch.cuda.current_stream().wait_event(ev)      # (the NEXT step waits for this copy: not a step ahead -- the worst case)
    return f


def h2d_ahead(stream, with_cast, new=False):
    """the copy for step i+1 is queued before step i and awaited (event) before step i+1, as the feeder does"""
    g = h2d_cast(stream) if with_cast else h2d(stream)
    pend = [None]
    def f():
        if pend[0] is not None:
            torch.cuda.current_stream().wait_event(pend[0])
        if new:
            new_fts()
        g()
        ev = torch.cuda.Event(); ev.record(stream); pend[0] = ev
    return f


def h2d_ahead_hostsync(stream, then_cast):
    """the copy for step i+1 is queued before step i; before step i+1 the HOST waits for the copy stream (no event behind the copy), then
    (optionally) queues the cast + an event that the step waits for"""
    g, cst = h2d(stream), cast_only(stream)
    started = [False]
    def f():
        if started[0]:
            stream.synchronize()
            if then_cast:
                cst()
                ev = torch.cuda.Event(); ev.record(stream)
                torch.cuda.current_stream().wait_event(ev)
        g(); started[0] = True
    return f


run("no copy", lambda: None)
run("H2D a step ahead, the HOST waits for the copy stream", h2d_ahead_hostsync(copy, False))
run("H2D a step ahead, host wait, then cast + event", h2d_ahead_hostsync(copy, True))
run("H2D a step ahead + event wait", h2d_ahead(copy, False))
run("H2D a step ahead + event wait + the trainer's copy of a new tensor", h2d_ahead(copy, False, True))
run("H2D + bist_cast a step ahead + event wait", h2d_ahead(copy, True))
run("the feature tensor counted as new every step (the trainer's 103 MB device-to-device copy)", new_fts)
run("bist_cast 205 -> 103 MB per step on the copy stream, no H2D", cast_only(copy))
run("H2D + bist_cast on the copy stream", h2d_cast(copy))
run("205 MB H2D per step on the feeder's copy stream", h2d(copy))
run("... in 8 pieces", h2d(copy, 8))
run("... on a pool stream", h2d(pool))
run("... + cast to bf16 on the copy stream", h2d(copy, 1, True))
run("no copy", lambda: None)
