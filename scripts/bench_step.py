"""Step time of the replayed training step alone (development aid): python scripts/bench_step.py [T] ; BIST_SPLIT_GRAPH=0/1."""
import os, sys, time
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
c = dict(bench.CFG)
if len(sys.argv) > 1:
    c["T"] = int(sys.argv[1])
from bist_amd import functional as Fn
if Fn.main_stream() is not None and os.environ.get("NULLSTREAM") != "1":
    torch.cuda.set_stream(Fn.main_stream())
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
for _ in range(5): tr.step(b)
torch.cuda.synchronize()
host = []
for _ in range(10):
    torch.cuda.synchronize(); h0 = time.perf_counter(); tr.step(b); host.append((time.perf_counter() - h0) * 1e3)
print("host ms per step() call with the device idle:", [round(h, 2) for h in host], flush=True)
res = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): terms = tr.step(b)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) * 50)
sp = tr._split
if os.environ.get("SPIN"):
    # what does ONE resident wave on a 4th hardware queue cost the step?  (bist_dev_idle_wave modes, 40 ms each)
    from bist_amd import graphsplit as GS2
    from bist_amd._lib import lib as L, check as CK
    extra = GS2.distinct_streams(4)[3]
    w = torch.zeros(4, dtype=torch.int64, device="cuda")
    for label, mode in (("no resident wave", -1), ("sleep + clock", 0), ("+ relaxed polls", 1), ("+ acquire polls", 2), ("sleep only", 3), ("no resident wave", -1),
                        ("torch._sleep", 9)):
        torch.cuda.synchronize()
        if mode == 9:
            with torch.cuda.stream(extra):
                torch.cuda._sleep(int(40e-3 * 2.0e9))
        elif mode >= 0:
            CK(L.bist_dev_idle_wave(extra.cuda_stream, int(40e-3 * 1e8) if mode < 3 else 12000, mode, w.data_ptr()), "idle")
        t0 = time.perf_counter()
        for _ in range(3): tr.step(b)
        torch.cuda.current_stream().synchronize()
        print(f"{label}: {(time.perf_counter() - t0) * 1e3 / 3:.3f} ms/step", flush=True)
        torch.cuda.synchronize()
from bist_amd import graphsplit as GS
print("usable:", GS._USABLE, "|", GS.WHY_NOT, flush=True)
print(f"T={c['T']} split={'on ' + str(sp.info) if sp is not None else 'off'}: ms/step {[round(r, 3) for r in res]}; loss {float(terms['out']):.4f}; "
      f"timed-out waits {sp.errors() if sp is not None else '-'}", flush=True)
if sp is not None and os.environ.get("DUMP"):
    sp.dump(os.environ["DUMP"])
if sp is not None and os.environ.get("TIMELINE"):
    tr.step(b)
    rows = sp.timeline()
    blocked = sorted(((e - b_, c_, k, near, fl, b_) for c_, k, near, fl, b_, e in rows if k == "wait"), reverse=True)
    print("span %.1f us; waits blocked longest:" % (max(r[5] for r in rows) - min(r[4] for r in rows)))
    for d, c_, k, near, fl, b_ in blocked[:40]:
        print(f"   chain {c_} before node {near:5d}: blocked {d:8.1f} us from t = {b_:8.1f} us (flags {fl})")
    tot = {}
    for d, c_, *_ in blocked: tot[c_] = tot.get(c_, 0.0) + d
    print("   blocked in total per chain (us):", {k: round(v, 1) for k, v in sorted(tot.items())})
