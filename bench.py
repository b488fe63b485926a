#!/usr/bin/env python3
"""Benchmark of the BiST hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one training step of the reference's loop (train.py:29-37: forward, pointer-generator
log-probs, 4 label-smoothed losses, backward, Adam with the Noam rate) on one synthetic batch of
BASELINE.json configs[1]: L=6, d_model=512, h=8, ResNeXt features [B=16, T=32, 7x7, 2048] bf16 already
resident in HBM, 20-token queries.  With N GPUs every rank runs its own B=16 batch (weak scaling,
batch-data-parallel), gradients are summed by RCCL all-reduces of the flat gradient buffer; the loss terms are
normalised by the token counts of the WHOLE batch (one all-reduce of two integers), as the reference does.
`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a child process,
before this process touches the GPU).

value       = target tokens (non-pad trg_y, the quantity train.py:36 accumulates) of all ranks / second.
roofline    = the north_star's region: the fused BiST attention forward (F_P0 + F_VL of SURVEY.md 8d: the video input
              projection + LayerNorm and ONE VidEncoderLayer4, both directions) at B=64, T=32, 7x7, C=2048 in inference
              mode: algorithmic FLOPs of the reference formulation / time of one replay of the region's hipGraph (HIP
              events on the launch stream), against the dense bf16 MFMA peak.  `kernels` lists the region's three
              dominant kernels (P0 GEMM, the two fused stage-1 launches) timed one by one with HIP events on their
              launch streams; `traffic` is the region's HBM bytes from the committed rocprofv3 --pmc passes.
cpu_baseline= the CPU oracle (oracle/bist_oracle.py, a port of the reference verified against it) on the host cores,
              rank 0, N=1 only, BASELINE.md section 2 protocol: full training step (forward + losses + backward + Adam)
              at B=16 in target tokens/s, 2 warm-up + 5 timed, median; `rows` adds the eval forwards and a decode turn.
"""
import argparse
import os

# HIP runtime switches the split-graph executor needs (bist_amd/__init__.py), set before anything can initialise the runtime
if os.environ.get("BIST_SPLIT_GRAPH", "1") != "0":
    os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
CFG = dict(L=6, d=512, h=8, B=16, T=32, S=49, C=2048, Lq=20, Lh=60, Lc=25, Lt=20, V=3000)


REGION_SOURCES = ("common.hpp", "gemm.hip", "rowops.hip", "attention.hip", "attention_mfma.hip", "st1_fused.hip")
REGION_PMC = "r04_attn_fwd_B64_pmc.json"


def region_sources_sha():
    """sha256 over the sources of the kernels of the roofline region (P0 GEMM, LayerNorm, the fused stage-1 launches, stage 2, the
    small products): the key of the committed counter passes -- traffic measured on other kernels is not this build's traffic."""
    import hashlib
    h = hashlib.sha256()
    for name in REGION_SOURCES:
        with open(os.path.join(ROOT, "bist_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def region_traffic(B, T):
    """HBM bytes of one pass over the roofline region from the committed PMC passes (profiles/r03_attn_fwd_B64_pmc.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of scripts/prof_attn.py, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md).  Counters cannot be read from inside a timed run, so this is the offline figure of the same kernels and
    shapes: (None, reason) when the geometry differs from the profiled one or the region's kernel sources changed since the passes
    (the file carries their sha256, scripts/pmc_region.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", REGION_PMC)) as f:
            t = json.load(f)
    except OSError:
        return None, "profiles/%s is missing" % REGION_PMC
    if t.get("B") != B or t.get("T") != T:
        return None, "the committed counter passes are of B=%s T=%s" % (t.get("B"), t.get("T"))
    if t.get("region_sources_sha256") != region_sources_sha():
        return None, "the region's kernel sources changed since the committed counter passes (re-run scripts/prof_round_r04.sh)"
    return t, None


def model_args(L, d, h, dropout):
    return argparse.Namespace(d_model=d, att_h=h, nb_blocks=L, nb_venc_blocks=L, nb_cenc_blocks=L, nb_aenc_blocks=0,
                              t2s=1, s2t=1, ptr_gen=1, ptr_ft="query,cap", mask_unk=1, auto_encoder=1,
                              include_caption="summary", enc_st_combine="none", dec_st_combine="seq",
                              enc_vc_combine="dyn", dropout=dropout, d_ff=4 * d)


def flops_alg(B, T, S, C, d, Lq, h):
    """Algorithmic FLOPs of SURVEY.md 8(d): (F_P0, F_VL) for one forward."""
    f_p0 = 2.0 * B * T * S * C * d
    a1 = 4 * B * S * T * d * d + 2 * B * Lq * d * d + 4 * B * S * Lq * T * d + 2 * B * S * Lq * d * d
    a2 = 4 * B * Lq * S * d * d + 4 * B * Lq * S * d + 4 * B * Lq * d * d
    a4 = 4 * B * T * S * d * d + 2 * B * Lq * d * d + 4 * B * T * Lq * S * d + 2 * B * T * Lq * d * d
    a5 = 4 * B * Lq * T * d * d + 4 * B * Lq * T * d + 4 * B * Lq * d * d
    a03 = 2 * (8 * B * Lq * d * d + 4 * B * Lq * Lq * d)
    ff = 2 * (16 * B * Lq * d * d)
    return f_p0, float(a1 + a2 + a4 + a5 + a03 + ff)


def flops_stage1(B, T, S, d, Lq, direction):
    """Reference-formulation FLOPs of stage 1 of one direction (the A1 / A4 terms above)."""
    G, K = (S, T) if direction == 0 else (T, S)
    return float(4 * B * G * K * d * d + 2 * B * Lq * d * d + 4 * B * G * Lq * K * d + 2 * B * G * Lq * d * d)


def cpu_baseline(rows_wanted, threads):
    """The oracle (port of the reference) on the host cores, BASELINE.md section 2: fp32, fixed seeds, 2 warm-up + 5 timed
    iterations, median (fewer iterations for the big rows, stated per row).  Threads: the CPU share of a one-GPU box (16 by
    default; on the 2 x 64-core host of the MI355X boxes the step takes 2.8 s with 16 threads, 3.2 s with 32, 15 s with 128)."""
    from oracle import bist_oracle as O
    c = CFG
    torch.set_num_threads(threads)
    cfg = O.Cfg(d_model=c["d"], att_h=c["h"], nb_blocks=c["L"], nb_venc_blocks=c["L"], nb_cenc_blocks=c["L"])
    torch.manual_seed(1234)
    sd = {}
    for k, s in O.state_shapes(cfg, c["V"], c["C"]).items():
        t = torch.randn(s) * (0.02 if len(s) > 1 else 0.01)
        if k.endswith(".a_2"):
            t = 1 + t
        sd[k] = t.requires_grad_(True)
    for alias in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj"):
        sd[alias] = sd["query_embed.0.lut.weight"]
    params = list({id(v): v for v in sd.values()}.values())
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9)          # train.py:129-130

    def batch(B, T, seed=1234):
        return O.det_batch(B, T, c["S"], c["C"], c["Lq"], c["Lh"], c["Lc"], c["Lt"], c["V"], seed=seed)

    def med(fn, warm, n):
        ts = []
        for it in range(warm + n):
            t0 = time.perf_counter()
            fn()
            if it >= warm:
                ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    ob = batch(c["B"], c["T"])

    def train_step():
        ft = O.mtn_forward(sd, cfg, ob)
        O.loss_compute(sd, cfg, ft, ob, c["V"])["total"].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    s_step = med(train_step, 2, 5)
    out = {"value": float(ob.ntokens) / s_step, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle (fp32, dropout off) full training step: forward + 4 losses + backward + torch Adam, B={c['B']} clips, "
                     f"T={c['T']}, L={c['L']}, 2 warm-up + 5 timed steps, median",
           "s_per_step": s_step, "nproc": os.cpu_count()}
    if rows_wanted == "all":
        rows = {}
        with torch.no_grad():
            rows["eval_forward_B16_T32_s"] = med(lambda: O.mtn_forward(sd, cfg, ob), 2, 5)
            ob64 = batch(64, c["T"])
            rows["eval_forward_B64_T32_s"] = med(lambda: O.mtn_forward(sd, cfg, ob64), 1, 3)
            del ob64
            ob128 = batch(c["B"], 128)
            rows["eval_forward_B16_T128_s"] = med(lambda: O.mtn_forward(sd, cfg, ob128), 1, 3)
            del ob128
            ob1 = batch(1, c["T"], seed=99)
            rows["decode_turn_beam5_maxlen12_s"] = med(lambda: O.beam_search(sd, cfg, ob1, 12, beam=5), 0, 1)
        rows["protocol"] = "median; B16: 2 warm-up + 5 timed, B64 / T128: 1 + 3, decode turn (60 re-evaluations of the reasoning layers, decode.py:59-66): 1 run"
        out["rows"] = rows
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child (this process has not
    touched the GPU) and leave with its exit code; rank 0 of the child writes the JSON line to our stdout."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=CFG["B"], help="clips per GPU")
    ap.add_argument("--T", type=int, default=CFG["T"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (the CPU share of a one-GPU box)")
    ap.add_argument("--cpu-rows", default="all", choices=["all", "step"], help="cpu_baseline: every BASELINE.md row, or the training step only")
    ap.add_argument("--no-decode", action="store_true", help="skip the beam-search turn timing (BASELINE configs[4])")
    ap.add_argument("--no-t128", action="store_true", help="skip the T=128 (BASELINE configs[3]) step / forward timing")
    ap.add_argument("--no-f32", action="store_true", help="skip the float32 side line (the same step in the parity dtype)")
    ap.add_argument("--no-fed", action="store_true", help="skip the PCIe-inclusive side line (features from host memory every step)")
    ap.add_argument("--no-graph", action="store_true", help="launch the kernels of a step eagerly instead of replaying a hipGraph")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout of rank 0 when its
    # communicator is created, so file descriptor 1 is pointed at stderr for the whole run and the result line is written
    # to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"bench.py --gpus {a.gpus} inside a launcher of {world} ranks")
    # rehearsal aid (one-GPU box): BIST_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges through gloo, so the
    # multi-rank control flow (barriers, gradient all-reduce, max-over-ranks timing) can be run without a second GPU
    rehearsal = os.environ.get("BIST_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif os.environ.get("BIST_BENCH_REHEARSAL") == "rccl1":
        # rehearsal aid (one-GPU box): a ONE-rank RCCL process group with the multi-rank exchange path forced on, so that RCCL
        # initialisation, the asynchronous bf16 all-reduces of the gradient pieces and their stream ordering run next to
        # the hipGraph capture / replay of the step in one process
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["BIST_FORCE_EXCHANGE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))

    import bist_amd.model as M
    from bist_amd import _lib, functional as Fn, ops
    from bist_amd.data.synthetic import synthetic_batch
    from bist_amd.train import Trainer

    # Everything below runs on the package's main stream instead of the NULL stream: the split-graph executor replays a graph's main chain
    # there, so a replay needs no event hop to and from the caller's stream -- hops that, from the NULL stream, each cost the runtime a walk
    # over all streams of the process (measured: the 0.75 ms region graph replays at 1.06 ms from the NULL stream once the process holds
    # the training and decode graphs' streams).  None: graphs go to the runtime's executor (BIST_SPLIT_GRAPH=0 / self-test failed).
    if not a.no_graph and Fn.main_stream() is not None:
        torch.cuda.set_stream(Fn.main_stream())
    c = dict(CFG, B=a.batch, T=a.T)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    args = model_args(c["L"], c["d"], c["h"], a.dropout)

    def make_batch(B, T, seed, Lh=None):
        return synthetic_batch(B, T=T, S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"] if Lh is None else Lh, Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=seed,
                               dtype=dtype)

    torch.manual_seed(1)                         # identical initial weights on every rank
    model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda()
    model.train()
    Fn.manual_seed(1234 + rank)
    trainer = Trainer(model, args, c["V"], compute_dtype=dtype, use_graph=not a.no_graph)
    batch = make_batch(c["B"], c["T"], 1234 + rank)
    ntok = int(batch.ntokens.item())

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_steps(tr, bt, warm, steps):
        for _ in range(warm):
            tr.step(bt)
        torch.cuda.synchronize(); barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(bt)
        torch.cuda.synchronize(); barrier()
        return time.perf_counter() - t0

    dt = timed_steps(trainer, batch, a.warmup, a.steps)
    # what replayed the step: the split-graph executor (one single-queue hipGraph per stream, device-flag syncs) or the runtime's
    sp = getattr(trainer, "_split", None)
    executor = {"kind": "split" if sp is not None else ("runtime" if trainer.use_graph else "eager")}
    if sp is not None:
        executor.update({k: sp.info[k] for k in ("chains", "nodes", "waits", "signals", "nodes_per_chain")})
        executor["timed_out_waits"] = sp.errors()
    hs, ready, whole = [], [], []
    for _ in range(10):                           # host time of one step() call with the device idle (hipGraphLaunch and the loop around it)
        torch.cuda.synchronize(); h0 = time.perf_counter(); trainer.step(batch); hs.append((time.perf_counter() - h0) * 1e3)
        ready.append(list(getattr(trainer, "exchange_ready_s", [])))
        torch.cuda.synchronize(); whole.append((time.perf_counter() - h0) * 1e3)
    executor["host_ms_per_step"] = sorted(hs)[len(hs) // 2]
    if getattr(trainer, "overlap", False):
        # several ranks (or the one-rank rehearsal): the buckets exchanged DURING the backward pass -- their sizes and when, after the
        # step's launch, the step's streams had signalled each as final (the host then issues its all-reduce); one isolated step
        es = trainer.flat_grad.element_size()
        executor["exchange"] = {"bucket_MB": [round((hi - lo) * es / 1e6, 1) for _, lo, hi in trainer.buckets],
                                "after_backward_MB": round((trainer._bucket_end - trainer.n32) * es / 1e6, 1),
                                "bucket_ready_ms": [round(sorted(r[j] for r in ready)[len(ready) // 2] * 1e3, 2) for j in range(len(trainer.buckets))],
                                "isolated_step_ms": round(sorted(whole)[len(whole) // 2], 2)}
        executor["host_ms_per_step"] = None       # (the host polls the buckets' flags inside step(): not a launch cost)
    tot = torch.tensor([dt, float(ntok)], device="cuda", dtype=torch.float64)
    if world > 1:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, ntok_all = float(tmax[0]), float(tsum[1])
    else:
        ntok_all = float(ntok)

    # ---- the roofline region: P0 + one VidEncoderLayer4, inference, replayed from a hipGraph ------------------------------------
    model.eval()

    def attn_forward_ms(bt):
        with torch.no_grad():
            q = model.encode_text(bt, {})["encoded_query"]
            vl = model.mutlimodal_decoder.v_layers[0]

            def run():
                f = model.vid_encoder(bt, {})
                vl({"t2s": q, "s2t": q}, f, bt)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    run()
            torch.cuda.current_stream().wait_stream(side)
            g = Fn.Graph()
            with Fn.capture_graph(g):
                run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            # the dominant kernels one by one: HIP events around each launch on its launch stream, three eager passes
            ops.GEMM_TIMING, ops.GEMM_TIMING_SHAPE, ops.ST1F_TIMING = [], (bt.fts.shape[0] * bt.fts.shape[1] * c["S"], c["d"], c["C"]), []
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            gt, st = ops.GEMM_TIMING, ops.ST1F_TIMING
            ops.GEMM_TIMING = ops.GEMM_TIMING_SHAPE = ops.ST1F_TIMING = None
            p0 = [e_0.elapsed_time(e_1) for _, e_0, e_1, _ in gt]
            plan = {pl for *_, pl in gt}
            st1 = {d_: [e_0.elapsed_time(e_1) for (_, _, _, _, dd), e_0, e_1 in st if dd == d_] for d_ in (0, 1)}
            return ms, (sum(p0) / max(1, len(p0)), plan), {d_: sum(v) / max(1, len(v)) for d_, v in st1.items()}

    def region_report(B, T, bt):
        ms, (p0_ms, p0_plan), st1_ms = attn_forward_ms(bt)
        f_p0, f_vl = flops_alg(B, T, c["S"], c["C"], c["d"], c["Lq"], c["h"])
        tfl = (f_p0 + f_vl) / (ms * 1e-3) / 1e12
        p0_kernel = {4: "gemm_big_kernel<bf16> (256x256 tiles)", 1: "gemm_fast_kernel<bf16> (128x128 tiles)"}.get(
            next(iter(p0_plan)) if len(p0_plan) == 1 else -1, "bist_gemm")
        kernels = [{"kernel": f"{p0_kernel} P0 [{B * T * c['S']}x{c['C']}]x[{c['C']}x{c['d']}]", "avg_launch_ms": p0_ms, "gflop_alg": f_p0 / 1e9,
                    "tflops": f_p0 / (p0_ms * 1e-3) / 1e12 if p0_ms > 0 else 0.0}]
        for d_, name in ((0, "t2s"), (1, "s2t")):
            fl = flops_stage1(B, T, c["S"], c["d"], c["Lq"], d_)
            kernels.append({"kernel": f"st1_fused_kernel {name} stage 1 (value projection + QK^T + softmax + PV + output projection + residual)",
                            "avg_launch_ms": st1_ms[d_], "gflop_alg": fl / 1e9, "tflops": fl / (st1_ms[d_] * 1e-3) / 1e12 if st1_ms[d_] > 0 else 0.0})
        return {"B": B, "T": T, "gflop_alg": (f_p0 + f_vl) / 1e9, "ms": ms, "tflops": tfl, "frac_of_mfma_peak": tfl / MFMA_BF16_PEAK_TFLOPS,
                "kernels": kernels}

    def decode_turn_ms():
        """BASELINE configs[4]: dialogue turns of beam_search_decode (B=1, beam 5, maxlen 12).  Two turns that may capture hipGraphs, then
        20 replayed turns, each bracketed by a device synchronisation: median, p90 and the per-turn list; what the turns did
        (bist_amd.model.decode.STATS: decided on the device / handed back to the host loop and run twice / captures) and the host's
        garbage collections during them are reported beside the times, so an outlier can be read.  Then the same without the per-turn
        reuse of the target-independent reasoning and late in a dialogue (200 history tokens)."""
        import gc
        from bist_amd.model import decode as D
        from bist_amd.model.decoder import MultimodalDecoder8
        b1 = make_batch(1, c["T"], 99)

        def turns(bt, capture, replay):
            ts = []
            for _ in range(capture):
                D.beam_search_decode(model, bt, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
            before, gc0 = dict(D.STATS), sum(g_["collections"] for g_ in gc.get_stats())
            for _ in range(replay):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                D.beam_search_decode(model, bt, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            did = {k: D.STATS[k] - before[k] for k in before}
            did["gc_collections"] = sum(g_["collections"] for g_ in gc.get_stats()) - gc0
            srt = sorted(ts)
            return {"median": srt[len(srt) // 2], "p90": srt[min(len(srt) - 1, int(0.9 * len(srt)))], "min": srt[0], "max": srt[-1],
                    "turns_ms": [round(t, 3) for t in ts], "did": did}
        with torch.no_grad():
            cached = turns(b1, 2, 20)
            MultimodalDecoder8.REASONING_CACHE = False
            try:
                recompute = turns(b1, 1, 2)
            finally:
                MultimodalDecoder8.REASONING_CACHE = True
            # the same turn late in a dialogue: 200 history tokens instead of configs[4]'s 60 (the decoder kernel's chunked attention core)
            h200 = turns(make_batch(1, c["T"], 99, Lh=200), 2, 10)
        return {"what": "beam_search_decode turn, B=1 beam=5 maxlen=12 (BASELINE configs[4]), one decode step at a time through the persistent decoder kernel, one hipGraph replay per (rows, position); "
                        "median of 20 replayed turns after 2 turns that capture",
                "ms_per_turn": cached["median"], "ms_per_turn_p90": cached["p90"], "ms_per_turn_min": cached["min"], "ms_per_turn_max": cached["max"],
                "turns_ms": cached["turns_ms"], "turns_did": cached["did"],
                "ms_per_turn_reasoning_recomputed": recompute["median"],
                "ms_per_turn_history_200_tokens": h200["median"], "history_200_tokens": {k: h200[k] for k in ("p90", "min", "max", "did")}}

    decode = decode_turn_ms() if (rank == 0 and not a.no_decode) else None
    attn = region_report(c["B"], c["T"], batch)
    attn64 = None
    if rank == 0 and c["B"] != 64:
        b64 = make_batch(64, c["T"], 4321)
        attn64 = region_report(64, c["T"], b64)
        del b64
    roof = attn64 if attn64 is not None else attn
    traffic, traffic_note = region_traffic(roof["B"], roof["T"]) if rank == 0 else (None, None)
    dom = max(roof["kernels"], key=lambda k: k["avg_launch_ms"])

    # BASELINE configs[3]: T=128 on the same model (its own trainer; rank 0 of a one-GPU run only)
    t128 = None
    if rank == 0 and world == 1 and not a.no_t128 and c["T"] != 128:
        del trainer
        torch.cuda.empty_cache()
        model.train()
        b128 = make_batch(c["B"], 128, 777)
        tr128 = Trainer(model, args, c["V"], compute_dtype=dtype, use_graph=not a.no_graph)
        dt128 = timed_steps(tr128, b128, 3, 5)
        model.eval()
        t128 = {"what": "BASELINE configs[3]: T=128 temporal segments, same model and batch size, training step and the roofline region",
                "ms_per_step": dt128 / 5 * 1e3, "tokens_per_s": float(b128.ntokens.item()) * 5 / dt128, "attn_fwd": region_report(c["B"], 128, b128)}
        del tr128, b128

    # the same step in float32 -- the dtype the parity tests hold to 1e-3 against the oracle; bf16 above is the throughput dtype
    f32 = None
    if rank == 0 and world == 1 and not a.no_f32 and a.dtype == "bf16":
        model.train()
        b32 = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234,
                              dtype=torch.float32)
        tr32 = Trainer(model, args, c["V"], compute_dtype=torch.float32, use_graph=not a.no_graph)
        dt32 = timed_steps(tr32, b32, 2, 4)
        model.eval()
        f32 = {"what": "the same training step with float32 operands and float32 kernels (the dtype of the 1e-3 parity tests); not the headline",
               "ms_per_step": dt32 / 4 * 1e3, "tokens_per_s": float(b32.ntokens.item()) * 4 / dt32}
        del tr32, b32

    # PCIe-inclusive side line: the same step with the features handed over as HOST fp32 tensors every step through DeviceFeeder
    # (pinned double buffer, H2D on a copy stream, cast on the device).  Never `value`.
    fed = None
    if rank == 0 and world == 1 and not a.no_fed and a.dtype == "bf16":
        from bist_amd.data.feeder import DeviceFeeder, HostBatch
        model.train()
        trf = Trainer(model, args, c["V"], compute_dtype=dtype, use_graph=not a.no_graph)
        host = []
        for i in range(2):
            hb = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234 + i,
                                 dtype=torch.float32, device="cpu")
            pf = DeviceFeeder.pinned_like(hb.fts.shape, hb.fts.dtype); pf.copy_(hb.fts)
            host.append(HostBatch(hb.query, hb.his, pf, hb.cap, hb.trg, hb.trg_y))
        tok = [int((h_.trg_y != 1).sum()) for h_ in host]          # non-pad target tokens per batch (pad id 1, dataset.py:98)
        nf = 20
        it = iter(DeviceFeeder([host[i % 2] for i in range(nf + 4)], feature_dtype=dtype))
        for _ in range(4):
            trf.step(next(it))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ntf, k = 0.0, 0
        for fb in it:
            trf.step(fb); ntf += tok[k % 2]; k += 1
        torch.cuda.synchronize(); dtf = time.perf_counter() - t0
        model.eval()
        fed = {"what": "the same training step with the features handed over as host fp32 tensors EVERY step (pinned producer buffers, DeviceFeeder: H2D a step "
                       "ahead on a copy stream that carries nothing but copies, cast to bf16 at the head of the step, temporal mask derived on the device); "
                       "PCIe-inclusive, not the headline",
               "ms_per_step": dtf / k * 1e3, "tokens_per_s": ntf / dtf, "h2d_bytes_per_step": int(host[0].fts.numel() * 4), "steps": k}
        del trf, it, host

    out = {
        "metric": "training-step tokens/sec (BiST hot path: fwd + pointer-generator losses + bwd + Adam)",
        "value": ntok_all * a.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 6-layer d_model=512 nhead=8 MTN, synthetic ResNeXt feats "
                               f"[B={c['B']}/GPU,T={c['T']},7x7,2048] {a.dtype} + 20-token queries, dropout={a.dropout}",
                   "global_batch": c["B"] * world, "tokens_per_step": ntok_all, "parallelism": f"dp{world}",
                   "clips_per_s": c["B"] * world * a.steps / dt},
        "roofline": {"bound": "mfma", "achieved": roof["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": roof["frac_of_mfma_peak"], "traffic": (traffic or {}).get("traffic_bytes"),
                     "traffic_detail": traffic if traffic is not None else {"note": traffic_note},
                     "region": f"fused BiST attention forward F_P0+F_VL (SURVEY 8d): P0 + LayerNorm + one VidEncoderLayer4, inference, "
                               f"B={roof['B']}, T={roof['T']}, 7x7, C=2048, one hipGraph replay",
                     "gflop_alg": roof["gflop_alg"], "avg_launch_ms": roof["ms"], "kernel": dom["kernel"], "kernels": roof["kernels"]},
        "attn_fwd": {"what": "the same region on the bench batch", **attn, "at_B64": attn64},
        "executor": executor,
        "decode": decode,
        "t128": t128,
        "f32": f32,
        "fed": fed,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_rows, a.cpu_threads)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        barrier()                              # rank 0 also ran the decode / B=64 extras: leave together
        dist.destroy_process_group()
    elif os.environ.get("BIST_BENCH_REHEARSAL") == "rccl1":
        import torch.distributed as dist1
        dist1.destroy_process_group()


if __name__ == "__main__":
    main()
