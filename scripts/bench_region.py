"""The roofline region (P0 + LayerNorm + one VidEncoderLayer4, inference, B=64) alone: replay time per executor / schedule (development aid)."""
import os, sys, time
if os.environ.get("BIST_SPLIT_GRAPH", "1") != "0":
    os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.synthetic import synthetic_batch
c = dict(bench.CFG)
B = int(os.environ.get("B", "64"))
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
bt = synthetic_batch(B, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=4321, dtype=torch.bfloat16)
with torch.no_grad():
    q = model.encode_text(bt, {})["encoded_query"]
    vl = model.mutlimodal_decoder.v_layers[0]

    def run():
        f = model.vid_encoder(bt, {})
        vl({"t2s": q, "s2t": q}, f, bt)
    for sched in [int(v) for v in os.environ.get("SCHEDS", "1,0,2").split(",")]:
        Fn.EVAL_SCHED = sched
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): run()
        torch.cuda.current_stream().wait_stream(side)
        g = Fn.Graph()
        with Fn.capture_graph(g):
            run()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        res = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g.replay()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20)
        one = []
        for rep in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            one.append(e0.elapsed_time(e1))
        info = g.split.info if g.split is not None else f"runtime executor, {g.streams} streams"
        print(f"B={B} sched {sched}: {[round(r, 4) for r in res]} ms per replay back to back; single replays {[round(r, 4) for r in one]}; {info}", flush=True)
        if g.split is not None and os.environ.get("TIMELINE"):
            for c_, k, near, fl, b_, e in g.split.timeline():
                print(f"     chain {c_} {k:6s} near node {near:3d} flags {fl}  begin {b_:9.1f} us  blocked {e - b_:9.1f} us", flush=True)
