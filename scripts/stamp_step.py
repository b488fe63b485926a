"""Device-side timeline of one REPLAYED training step (development aid, round 4): bist_amd/stamps.py queues one-thread timestamp launches at
named points of the three streams, forward and backward; captured with the step, a replay fills them.  Prints, per stream, the points in
time order with the gap to the previous point of that stream -- the chain of dependent launches that bounds the step shows as the
longest path through these columns.  (Each stamp is a launch of its own: ~2 us on its chain, ~150 of them per step.)

    python scripts/stamp_step.py [T]        ->  stdout (commit as profiles/r04_step_stamps*.txt)
"""
import os, sys
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import stamps as S
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer

c = dict(bench.CFG)
if len(sys.argv) > 1:
    c["T"] = int(sys.argv[1])
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
S.enable()
for _ in range(5):
    tr.step(b)           # the first call captures (two eager warm-up passes + the capture: only the capture's slots are refilled by replays)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    tr.step(b)
torch.cuda.synchronize()
print(f"stamped step: {(time.perf_counter() - t0) * 50:.3f} ms per step over 20 replays ({len(S.NAMES)} stamp launches recorded over warm-up + capture)")
S.BUF.zero_()
torch.cuda.synchronize()
if os.environ.get("IDLE"):
    from bist_amd import graphsplit as GS2
    from bist_amd._lib import lib as L, check as CK
    extra = GS2.distinct_streams(4)[3]
    wq = torch.zeros(4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    CK(L.bist_dev_idle_wave(extra.cuda_stream, int(150e-3 * 1e8), 0, wq.data_ptr()), "idle")
    for _ in range(int(os.environ.get("IDLE_PENDING", "0"))):      # launches queued BEHIND the resident wave on its queue
        CK(L.bist_dev_idle_wave(extra.cuda_stream, 10, 0, wq.data_ptr()), "idle")
if os.environ.get("STEADY", "1") != "0":
    for _ in range(8):        # the stamps of the LAST of several back-to-back replays: the steady state the bench measures
        tr.step(b)
else:
    tr.step(b)
rows = S.read()
streams = sorted({s for _, s, _ in rows})
main = next(s for n, s, _ in rows if n == "step head")          # (the capture's own stream)
cap = next((s for n, s, _ in rows if "cap in" in n), None)
label = {s: ("main" if s == main else "cap/dec" if s == cap else "s2t") for s in streams}
print(f"{len(rows)} stamps in the replay; span {rows[-1][2] - rows[0][2]:.1f} us; streams: " + ", ".join(f"{label[s]}={s:#x}" for s in streams))
last = {}
print("%10s  %-6s %9s  %s" % ("t (us)", "stream", "gap (us)", "point"))
for name, s, t in rows:
    gap = t - last[s] if s in last else 0.0
    last[s] = t
    print("%10.1f  %-6s %9.1f  %s" % (t, label[s], gap, name))
