"""Soak of the captured-graph paths: many capture + replay cycles over small models, interleaved with allocations and
collector runs -- the conditions under which replays of the runtime's multi-branch graph executor crashed in round 3.

    python scripts/soak_graphs.py [split|runtime] [captures] [replays]

Trainers: default (optimiser in the step), deferred optimiser, the two-graph step of the several-rank path (one-rank exchange forced,
gloo), fp32 and bf16; decode: beam search over several dialogue geometries (first-step graphs, step graphs).  Prints one summary line
and exits 0; any crash is this process's exit code (tests/test_soak_gpu.py runs it as a child for exactly that reason)."""
import argparse, gc, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
n_cap = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
if mode == "split":
    os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
else:
    os.environ["BIST_SPLIT_GRAPH"] = "0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from oracle import bist_oracle as O
import bist_amd.model as M
from bist_amd import functional as Fn, graphsplit as GS
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.model import decode as D
from bist_amd.train import Trainer

cfg = O.Cfg(d_model=64, att_h=4, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
args = argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})
torch.manual_seed(0)
t0 = time.time()
caps = reps = 0
junk = []
junk_graphs = []
max_streams = 0


def churn(i):
    if os.environ.get("SOAK_NOCHURN"):
        return
    _churn(i)


def _churn(i):
    """allocations of changing sizes + a collector run now and then: cyclic garbage owning graphs and buffers dies HERE, not inside a capture"""
    junk.append(torch.empty((1 + (i * 7919) % 4096, 33), device="cuda"))
    if len(junk) > 12:
        del junk[:6]
    if i % 7 == 0:
        gc.collect()


kinds = [("default", dict(), torch.float32), ("default", dict(), torch.bfloat16), ("deferred", dict(deferred_adam=True), torch.float32),
         ("deferred", dict(deferred_adam=True), torch.bfloat16)]
if os.environ.get("SOAK_KINDS"):          # development aid: restrict the trainer kinds ("default", "deferred") / dtypes ("f32", "bf16")
    sel = os.environ["SOAK_KINDS"].split(",")
    kinds = [k for k in kinds if k[0] in sel and (("f32" in sel) == (k[2] == torch.float32) or not ({"f32", "bf16"} & set(sel)))]
batches = {dt: [synthetic_batch(3, T=5, S=9, C=64, Lq=6, Lh=8, Lc=5, Lt=5, vocab=80, seed=s_, dtype=dt) for s_ in (1, 2)] for dt in (torch.float32, torch.bfloat16)}
batches2 = {dt: [synthetic_batch(2, T=4, S=9, C=64, Lq=5, Lh=7, Lc=4, Lt=6, vocab=80, seed=s_, dtype=dt) for s_ in (3, 4)] for dt in (torch.float32, torch.bfloat16)}
per_trainer_caps = 4
rounds = max(1, (n_cap * 2 // 3) // (len(kinds) * per_trainer_caps))
replays_each = max(2, (n_rep * 2 // 3) // (rounds * len(kinds) * per_trainer_caps))
for r in range(rounds):
    for tag, kw, dt in kinds:
        model = M.make_model(80, 80, args, ft_sizes=[64]).cuda()
        model.train()
        tr = Trainer(model, args, 80, compute_dtype=dt, warmup=20, factor=2.0, use_graph=True, **kw)
        for c in range(per_trainer_caps):           # a new batch geometry forces a new capture
            bs = (batches if c % 2 == 0 else batches2)[dt]
            if os.environ.get("SOAK_LEAK"):          # development aid: retire the old graphs instead of destroying them
                junk_graphs.append((tr._graph, tr._graph2))
            if os.environ.get("SOAK_SYNC"):
                torch.cuda.synchronize()
            tr._graph = None
            for k in range(replays_each):
                if os.environ.get("SOAK_TRACE"):
                    print("step", r, tag, dt, c, k, file=sys.stderr, flush=True)
                out = tr.step(bs[k % 2])
                reps += 1
                churn(reps)
            caps += 1
            max_streams = max(max_streams, Fn.capture_graph.LAST_STREAMS)
            if tr._split is not None:
                assert tr._split.errors() == 0, "a wait of the split executor timed out"
        loss = float(out["out"])
        assert loss == loss, "NaN loss"
        model.eval()
        del tr, model
# the two-graph step of the several-rank path on one rank (gloo group of one)
if not dist.is_initialized():
    dist.init_process_group("gloo", rank=0, world_size=1)
os.environ["BIST_FORCE_EXCHANGE"] = "1"
for r in range(max(1, rounds // 2)):
    for dt in (torch.float32, torch.bfloat16):
        model = M.make_model(80, 80, args, ft_sizes=[64]).cuda(); model.train()
        tr = Trainer(model, args, 80, compute_dtype=dt, warmup=20, factor=2.0, use_graph=True)
        assert tr.exchanging
        for c in range(2):
            if os.environ.get("SOAK_LEAK"):          # development aid: retire the old graphs instead of destroying them
                junk_graphs.append((tr._graph, tr._graph2))
            if os.environ.get("SOAK_SYNC"):
                torch.cuda.synchronize()
            tr._graph = None
            for k in range(replays_each):
                tr.step((batches if c % 2 == 0 else batches2)[dt][k % 2]); reps += 1; churn(reps)
            caps += 2
        model.eval(); del tr, model
os.environ.pop("BIST_FORCE_EXCHANGE")
# decode: several dialogue geometries, graphs dropped and re-captured
dmodel = M.make_model(80, 80, args, ft_sizes=[64]).cuda().eval()
geoms = [(6, 8, 5), (9, 20, 7), (6, 40, 5), (12, 70, 9)]
turns = 0
want_caps = max(1, n_cap // 3)
with torch.no_grad():
    ref = {}
    while D.STATS["captures"] < want_caps or turns < n_rep // 12:
        for gi, (Lq, Lh, Lc) in enumerate(geoms):
            b1 = synthetic_batch(1, T=5, S=9, C=64, Lq=Lq, Lh=Lh, Lc=Lc, Lt=5, vocab=80, seed=50 + gi, dtype=torch.float32)
            res = D.beam_search_decode(dmodel, b1, 8, 2, 0, 3, 1, beam=3, penalty=1.0, nbest=3, train_args=args)
            toks = [r_[0] for r_ in res[0]]
            assert ref.setdefault(gi, toks) == toks, "a replayed turn decoded differently"
            turns += 1
            churn(turns)
        if turns % 24 == 0:
            D._drop_graphs(dmodel)                    # force re-captures
        if turns > 20000:
            break
caps += D.STATS["captures"]
torch.cuda.synchronize()
print(f"soak ok: mode {mode}, split usable {GS._USABLE}, {caps} captures, {reps} trainer replays, {turns} decode turns ({D.STATS['captures']} decode captures), "
      f"widest capture {max_streams} streams, {time.time() - t0:.0f} s", flush=True)
