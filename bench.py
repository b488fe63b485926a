#!/usr/bin/env python3
"""Benchmark of the BiST hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one training step of the reference's loop (train.py:29-37: forward, pointer-generator
log-probs, 4 label-smoothed losses, backward, Adam with the Noam rate) on one synthetic batch of
BASELINE.json configs[1]: L=6, d_model=512, h=8, ResNeXt features [B=16, T=32, 7x7, 2048] bf16 already
resident in HBM, 20-token queries.  With N GPUs every rank runs its own B=16 batch (weak scaling,
batch-data-parallel), gradients are summed by one RCCL all-reduce of the flat gradient buffer.

value       = target tokens (non-pad trg_y, the quantity train.py:36 accumulates) of all ranks / second.
roofline    = the dominant kernel by FLOPs, the video input projection GEMM (P0: [B*T*S,2048]x[2048,512],
              bist_gemm LDS-DMA MFMA kernel): algorithmic 2*M*N*K FLOPs / its mean launch duration,
              measured with HIP events on the launch stream inside the timed steps, against the dense
              bf16 MFMA peak.  `attn_fwd` adds the north_star's second figure: the fused BiST attention
              forward (F_P0 + F_VL of SURVEY.md 8d) timed on the same batch.
cpu_baseline= the CPU oracle's (oracle/bist_oracle.py, a port) training step (fwd + loss + backward; no
              optimiser) on a bounded sample of the same workload on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
CFG = dict(L=6, d=512, h=8, B=16, T=32, S=49, C=2048, Lq=20, Lh=60, Lc=25, Lt=20, V=3000)


def p0_traffic(M, K, N):
    """HBM bytes per P0 launch from the committed PMC passes (profiles/r01_p0_traffic_pmc.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this benchmark, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md); None when the geometry differs from the profiled one.  Counters cannot be read from
    inside a timed run, so this is the offline figure of the same kernel and shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_p0_traffic_pmc.json")) as f:
            t = json.load(f)
    except OSError:
        return None
    if t.get("algorithmic_bytes_per_launch") != (M * K + N * K + M * N) * 2:
        return None
    return {"bytes_per_launch": t["traffic_bytes_per_launch"], "algorithmic_bytes": t["algorithmic_bytes_per_launch"],
            "ratio": t["ratio"], "source": "profiles/r01_p0_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, offline)"}


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


def cpu_baseline(sample_B, steps, dropout):
    """The oracle (port of the reference) on the host cores: forward + losses + backward at the bench geometry."""
    from oracle import bist_oracle as O
    c = CFG
    cfg = O.Cfg(d_model=c["d"], att_h=c["h"], nb_blocks=c["L"], nb_venc_blocks=c["L"], nb_cenc_blocks=c["L"])
    torch.manual_seed(1)
    shapes = O.state_shapes(cfg, c["V"], c["C"])
    sd = {}
    for k, s in shapes.items():
        t = torch.randn(s) * (0.02 if len(s) > 1 else 0.01)
        if k.endswith(".a_2"):
            t = 1 + t
        sd[k] = t.requires_grad_(True)
    for alias in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj"):
        sd[alias] = sd["query_embed.0.lut.weight"]
    ob = O.det_batch(sample_B, c["T"], c["S"], c["C"], c["Lq"], c["Lh"], c["Lc"], c["Lt"], c["V"], seed=99)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        ft = O.mtn_forward(sd, cfg, ob)
        O.loss_compute(sd, cfg, ft, ob, c["V"])["total"].backward()
        for v in sd.values():
            v.grad = None
        if it > 0:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": float(ob.ntokens) / dt, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fwd+loss+bwd (no optimiser), B={sample_B} clips of the same geometry, median of {steps} steps, fp32",
            "s_per_step": dt}


def main():
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout of rank 0 when its
    # communicator is created, so file descriptor 1 is pointed at stderr for the whole run and the result line is written
    # to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=CFG["B"], help="clips per GPU")
    ap.add_argument("--T", type=int, default=CFG["T"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the beam-search turn timing (BASELINE configs[4])")
    ap.add_argument("--cpu-sample", type=int, default=2)
    ap.add_argument("--no-graph", action="store_true", help="launch the ~2.7k kernels of a step eagerly instead of replaying a hipGraph")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
    from bist_amd import functional as Fn, ops
    from bist_amd.data.synthetic import synthetic_batch
    from bist_amd.train import Trainer

    c = dict(CFG, B=a.batch, T=a.T)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    args = model_args(c["L"], c["d"], c["h"], a.dropout)
    torch.manual_seed(1)                         # identical initial weights on every rank
    model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda()
    model.train()
    Fn.manual_seed(1234 + rank)
    trainer = Trainer(model, args, c["V"], compute_dtype=dtype, use_graph=not a.no_graph)
    batch = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"],
                            seed=1234 + rank, dtype=dtype)
    ntok = int(batch.ntokens.item())

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        trainer.step(batch)
    torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step(batch)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    # Dominant-kernel timing: HIP events around every P0 GEMM launch of three further training passes run
    # eagerly (the timed steps replay a hipGraph, whose kernel nodes cannot carry timing events on ROCm:
    # "External events are disallowed in rocm"), same process, same batch, same stream.
    ops.GEMM_TIMING, ops.GEMM_TIMING_SHAPE = [], (c["B"] * c["T"] * c["S"], c["d"], c["C"])
    for _ in range(3):
        trainer.backward(batch)
    torch.cuda.synchronize()
    timing, ops.GEMM_TIMING, ops.GEMM_TIMING_SHAPE = ops.GEMM_TIMING, None, None

    tot = torch.tensor([dt, float(ntok)], device="cuda", dtype=torch.float64)
    if world > 1:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, ntok_all = float(tmax[0]), float(tsum[1])
    else:
        ntok_all = float(ntok)

    # dominant kernel: the P0 GEMM (M = B*T*S, N = d, K = C), forward launches only
    M_p0 = c["B"] * c["T"] * c["S"]
    p0 = [e0.elapsed_time(e1) for (m, n, k, z), e0, e1, _ in timing if (m, n, k, z) == (M_p0, c["d"], c["C"], 1)]
    p0_plan = {plan for (m, n, k, z), _, _, plan in timing if (m, n, k, z) == (M_p0, c["d"], c["C"], 1)}
    p0_kernel = {4: "gemm_big_kernel<bf16> (256x256 tiles)", 1: "gemm_fast_kernel<bf16> (128x128 tiles)"}.get(
        next(iter(p0_plan)) if len(p0_plan) == 1 else -1, "bist_gemm")
    p0_ms = sum(p0) / max(1, len(p0))
    p0_flops = 2.0 * M_p0 * c["d"] * c["C"]
    achieved = p0_flops / (p0_ms * 1e-3) / 1e12 if p0_ms > 0 else 0.0

    # fused BiST attention forward (F_P0 + F_VL): P0 + one VidEncoderLayer4, eval mode, replayed from a hipGraph
    # (the two directions run as parallel graph branches).  Measured on the bench batch (B=16) and at the
    # north_star's roofline point B=64 on a second synthetic batch.
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
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20

    def decode_turn_ms():
        """BASELINE configs[4]: one dialogue turn of beam_search_decode (B=1, beam 5, maxlen 12) with and without the
        per-turn reuse of the target-independent reasoning."""
        from bist_amd.model.decode import beam_search_decode
        from bist_amd.model.decoder import MultimodalDecoder8
        b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=dtype)
        res = {}
        with torch.no_grad():
            for name, flag in (("cached", True), ("recompute", False)):
                MultimodalDecoder8.REASONING_CACHE = flag
                try:
                    ts = []
                    for _ in range(3):
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
                        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
                    res[name] = min(ts)
                finally:
                    MultimodalDecoder8.REASONING_CACHE = True
        return {"what": "beam_search_decode turn, B=1 beam=5 maxlen=12 (BASELINE configs[4]), hipGraph replay per (rows, prefix length)", "ms_per_turn": res["cached"],
                "ms_per_turn_reasoning_recomputed": res["recompute"]}

    decode = decode_turn_ms() if (rank == 0 and not a.no_decode) else None
    attn_ms = attn_forward_ms(batch)
    f_p0, f_vl = flops_alg(c["B"], c["T"], c["S"], c["C"], c["d"], c["Lq"], c["h"])
    attn_tflops = (f_p0 + f_vl) / (attn_ms * 1e-3) / 1e12
    attn64 = None
    if rank == 0 and c["B"] != 64:
        b64 = synthetic_batch(64, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"],
                              seed=4321, dtype=dtype)
        ms64 = attn_forward_ms(b64)
        p64, v64 = flops_alg(64, c["T"], c["S"], c["C"], c["d"], c["Lq"], c["h"])
        attn64 = {"B": 64, "gflop_alg": (p64 + v64) / 1e9, "ms": ms64, "tflops": (p64 + v64) / (ms64 * 1e-3) / 1e12,
                  "frac_of_mfma_peak": (p64 + v64) / (ms64 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        del b64

    out = {
        "metric": "training-step tokens/sec (BiST hot path: fwd + pointer-generator losses + bwd + Adam)",
        "value": ntok_all * a.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 6-layer d_model=512 nhead=8 MTN, synthetic ResNeXt feats "
                               f"[B={c['B']}/GPU,T={c['T']},7x7,2048] {a.dtype} + 20-token queries, dropout={a.dropout}",
                   "global_batch": c["B"] * world, "tokens_per_step": ntok_all, "parallelism": f"dp{world}",
                   "clips_per_s": c["B"] * world * a.steps / dt},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": (p0_traffic(M_p0, c["C"], c["d"]) or {}).get("bytes_per_launch"),
                     "traffic_detail": p0_traffic(M_p0, c["C"], c["d"]),
                     "kernel": f"{p0_kernel} P0 [{M_p0}x{c['C']}]x[{c['C']}x{c['d']}]", "avg_launch_ms": p0_ms,
                     "launches_timed": len(p0)},
        "attn_fwd": {"what": "fused BiST attention forward F_P0+F_VL (SURVEY 8d), one layer, eval, hipGraph replay", "B": c["B"], "gflop_alg": (f_p0 + f_vl) / 1e9,
                     "ms": attn_ms, "tflops": attn_tflops, "frac_of_mfma_peak": attn_tflops / MFMA_BF16_PEAK_TFLOPS,
                     "at_B64": attn64},
        "decode": decode,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_sample, 3, a.dropout)
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
