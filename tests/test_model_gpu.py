"""End-to-end parity of the HIP model (bist_amd.model, through the C ABI) against the CPU oracle and
the golden vectors captured from the reference.

Tolerances (north_star): fp32 path -- every ``ft`` tensor and the generator log-probs within 1e-3 of
the reference (absolute, on O(1) layer-normed activations), argmax-identical; bf16 path -- bf16
storage error, checked as <= 6e-2 absolute on layer-normed activations plus >= 90 % identical greedy
argmax (it is the throughput path, not the parity gate).
"""
import argparse
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import bist_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import bist_amd.model as M
    from bist_amd.data.batch import Batch
    return M, Batch


def _args(cfg: O.Cfg):
    return argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})


def build_model(M, cfg, V, C, dtype=torch.float32):
    torch.manual_seed(0)
    model = M.make_model(V, V, _args(cfg), ft_sizes=[C])
    sd = O.det_state(cfg, V, C)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(".pe") for k in missing), (missing, unexpected)
    assert model.generator.vocab_gen is model.query_embed[0].lut.weight      # sharing survives load
    model = model.to("cuda").to(dtype).eval()
    return model, sd


def to_batch(Batch, ob: O.OBatch, dtype=torch.float32):
    dev = "cuda"
    return Batch(ob.query.to(dev), ob.his.to(dev), ob.fts.to(dev).to(dtype), ob.cap.to(dev), ob.trg.to(dev),
                 ob.trg_y.to(dev) if ob.trg_y is not None else None)


def _err(a, b):
    return (a.detach().float().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def _setup(g, tag):
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"],
                     seed=dm.get("seed", 1234), fully_masked_clip=meta.get("fully_masked", False))
    return cfg, dm, ob


@pytest.mark.parametrize("tag", ["both", "mid", "t2s", "s2t", "masked"])
def test_forward_matches_reference_golden_fp32(hip, golden_dir, tag):
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g3_model.npz"))
    cfg, dm, ob = _setup(g, tag)
    model, sd = build_model(M, cfg, dm["V"], dm["C"])
    b = to_batch(Batch, ob)
    assert torch.equal(b.temporal_mask.cpu(), ob.temporal_mask)               # device-side mask == dataset.py:79
    with torch.no_grad():
        ft = model.forward(b)
        logp = model.generator(ft, b, _args(cfg))
    worst = {}
    for k in [k for k in g.files if k.startswith(f"{tag}_ft_")]:
        name = k[len(tag) + 4:]
        worst[name] = _err(ft[name], g[k])
    worst["logp"] = _err(logp, g[f"{tag}_logp"])
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, f"beyond 1e-3 of the reference: {bad} (all: {worst})"
    assert torch.equal(logp.argmax(-1).cpu(), torch.from_numpy(g[f"{tag}_logp"]).argmax(-1))
    # losses through the HIP loss kernels
    from bist_amd.model.label_smoothing import LabelSmoothing
    from bist_amd.model.optimize import SimpleLossCompute
    crit = LabelSmoothing(dm["V"], O.PAD_ID, 0.1)
    with torch.no_grad():
        terms, _ = SimpleLossCompute(model.generator, model.ae_generator, crit, None, args=_args(cfg)).terms(ft, b)
    for name, val in terms.items():  # device scalars
        ref = float(g[f"{tag}_loss_{name}"])
        assert abs(val.item() - ref) <= 1e-3 * max(1.0, abs(ref)), (name, val.item(), ref)


def test_forward_matches_oracle_config2_shape_fp32(hip):
    """d_model=512, h=8, C=2048, T=32, S=49, Lq=20 (BASELINE config 2 geometry) with 2 layers and B=2 so the
    CPU oracle finishes in seconds; every kernel runs at its production tile shapes."""
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 2048
    ob = O.det_batch(2, 32, 49, C, 20, 30, 15, 12, V, seed=5)
    model, sd = build_model(M, cfg, V, C)
    b = to_batch(Batch, ob)
    with torch.no_grad():
        ft = model.forward(b)
        logp = model.generator(ft, b, _args(cfg))
        ref = O.mtn_forward(sd, cfg, ob)
        ref_logp = O.multi_pointer_generator(sd, cfg, ref, ob)
    worst = {k: _err(ft[k], ref[k]) for k in ref}
    worst["logp"] = _err(logp, ref_logp)
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, f"beyond 1e-3 of the oracle: {bad} (all: {worst})"
    assert torch.equal(logp.argmax(-1).cpu(), ref_logp.argmax(-1))


def test_forward_bf16_close_to_oracle(hip, golden_dir):
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g3_model.npz"))
    cfg, dm, ob = _setup(g, "mid")
    model, sd = build_model(M, cfg, dm["V"], dm["C"], torch.bfloat16)
    b = to_batch(Batch, ob, torch.bfloat16)
    with torch.no_grad():
        ft = model.forward(b)
        logp = model.generator(ft, b, _args(cfg))
    for name in ("temporal_ft", "spatial_ft", "cap_ft", "encoded_ft", "decoded_text"):
        e = _err(ft[name], g[f"mid_ft_{name}"])
        assert e <= 6e-2, (name, e)
    agree = (logp.argmax(-1).cpu() == torch.from_numpy(g["mid_logp"]).argmax(-1)).float().mean().item()
    assert agree >= 0.9, agree


@pytest.mark.parametrize("tag", ["beam5", "beam1"])
def test_beam_search_matches_reference(hip, golden_dir, tag):
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g5_beam.npz"))
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    model, _ = build_model(M, cfg, dm["V"], dm["C"])
    b = to_batch(Batch, ob)
    from bist_amd.model.decode import beam_search_decode
    with torch.no_grad():
        hyps, best = beam_search_decode(model, b, dm["maxlen"], O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=meta["beam"],
                                        penalty=1.0, nbest=5, train_args=_args(cfg))
    assert len(hyps) == int(g[f"{tag}_n"])
    for i, (toks, score) in enumerate(hyps):
        assert [int(t) for t in toks] == g[f"{tag}_hyp{i}"].tolist(), f"hyp {i}"
        assert abs(float(score) - float(g[f"{tag}_score{i}"])) < 1e-3
    assert abs(float(best) - float(g[f"{tag}_best"])) < 1e-3
    # one model.decode per hypothesis (the reference's loop) gives the same n-best as the batched hypotheses
    import bist_amd.model.decode as D
    D.BATCH_HYPOTHESES = False
    try:
        with torch.no_grad():
            hyps2, best2 = beam_search_decode(model, to_batch(Batch, ob), dm["maxlen"], O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID,
                                              beam=meta["beam"], penalty=1.0, nbest=5, train_args=_args(cfg))
    finally:
        D.BATCH_HYPOTHESES = True
    assert [[int(t) for t in h[0]] for h in hyps2] == [[int(t) for t in h[0]] for h in hyps]
    assert all(abs(float(a[1]) - float(b[1])) < 1e-4 for a, b in zip(hyps, hyps2))


def test_decode_reuses_the_target_independent_reasoning(hip, golden_dir):
    """across the model.decode calls of one turn (decode.py:66) the visual / caption reasoning is
    computed once (kept in the turn's ``ft`` dict); results equal the recompute-every-call behaviour of the reference."""
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g5_beam.npz"))
    meta = json.loads(str(g["beam5_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    model, _ = build_model(M, cfg, dm["V"], dm["C"])
    b = to_batch(Batch, ob)
    dec = model.mutlimodal_decoder
    calls = []
    hook = dec.v_layers[0].register_forward_hook(lambda *a: calls.append(1))
    try:
        with torch.no_grad():
            ft = model.encode(b)
            outs = []
            for _ in range(3):
                ft = model.decode(b, ft)
                outs.append(ft["decoded_text"].clone())
            assert len(calls) == 1 and "_bist_reasoning" in ft, "the reasoning layers must run once per turn"
            type(dec).REASONING_CACHE = False
            try:
                ft2 = model.decode(b, model.encode(b))
            finally:
                type(dec).REASONING_CACHE = True
            assert len(calls) == 2
    finally:
        hook.remove()
    for o in outs:
        assert torch.equal(o, outs[0])
    assert _err(ft2["decoded_text"], outs[0].cpu().float().numpy()) < 1e-6
    for k in ("temporal_ft", "spatial_ft", "cap_ft", "encoded_ft"):
        assert torch.equal(ft[k], ft2[k]), k
    # training / autograd never uses the cache
    with torch.enable_grad():
        ft3 = model.encode(b)
        model.decode(b, ft3)
        assert "_bist_reasoning" not in ft3


def test_generic_module_signatures(hip):
    """MultiHeadedAttention / PositionwiseFeedForward / LayerNorm keep the reference call signatures."""
    M, _ = hip
    from bist_amd.model.modules import LayerNorm, MultiHeadedAttention, PositionwiseFeedForward
    d, h = 64, 4
    sd = {f"linears.{j}.weight": O.det_param(f"prim.attn.linears.{j}.weight", (d, d)) for j in range(4)}
    sd.update({f"linears.{j}.bias": O.det_param(f"prim.attn.linears.{j}.bias", (d,)) for j in range(4)})
    att = MultiHeadedAttention(h, d, dropout=0.0)
    att.load_state_dict(sd)
    att = att.cuda().eval()
    att.keep_attn = True
    g = torch.Generator().manual_seed(3)
    q, kv = torch.randn(3, 5, d, generator=g), torch.randn(3, 7, d, generator=g)
    mask = torch.ones(3, 1, 7, dtype=torch.bool); mask[1, 0, 4:] = False; mask[2] = False
    osd = {"a." + k: v for k, v in sd.items()}
    ref, pref = O.mha(osd, "a", h, q, kv, kv, mask)
    with torch.no_grad():
        out = att(q.cuda(), kv.cuda(), kv.cuda(), mask.cuda())
    assert _err(out, ref) < 1e-4 and _err(att.attn, pref) < 1e-5


@pytest.mark.parametrize("tag", ["T32", "T128"])
def test_full_size_fp32_matches_reference_digests(hip, golden_dir, tag):
    """the fp32 HIP path at d_model=512, L=6, T=32 / T=128 against digests of the REFERENCE's own outputs
    (tests/golden/g6_fullsize.npz, regenerated from seed): every ft tensor and the log-probs within 1e-3, argmax identical; at
    T=32 also the beam-5 n-best of a dialogue decoded with the full-size model."""
    from test_oracle_golden import _digest, check_digest
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g6_fullsize.npz"))
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    model, sd = build_model(M, cfg, dm["V"], dm["C"])
    b = to_batch(Batch, ob)
    with torch.no_grad():
        ft = model.forward(b)
        logp = model.generator(ft, b, _args(cfg))
    for k in [k for k in g.files if k.startswith(f"{tag}_dg_") and k != f"{tag}_dg_logp"]:
        name = k[len(tag) + 4:]
        check_digest(_digest(ft[name].float().cpu()), g[k], f"{tag} {name}", TOL)
    check_digest(_digest(logp.float().cpu()), g[f"{tag}_dg_logp"], f"{tag} logp", TOL)
    assert np.array_equal(logp.argmax(-1).cpu().numpy(), g[f"{tag}_argmax"])
    if meta.get("beam"):
        ob1 = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"] + 1)
        with torch.no_grad():
            hyps, _ = beam_search_decode(model, to_batch(Batch, ob1), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=meta["beam"],
                                         penalty=1.0, nbest=5, train_args=_args(cfg))
        assert len(hyps) == int(g[f"{tag}_beam_n"])
        for i, (toks, score) in enumerate(hyps):
            assert [int(t) for t in toks] == g[f"{tag}_beam_hyp{i}"].tolist(), i
            assert abs(float(score) - float(g[f"{tag}_beam_score{i}"])) <= 1e-3 * max(1.0, abs(float(g[f"{tag}_beam_score{i}"])))


@pytest.mark.parametrize("n,Lt,Lh", [(1, 1, 30), (3, 5, 30), (5, 12, 30), (2, 7, 30), (5, 1, 65), (3, 5, 100), (5, 12, 128), (2, 7, 129), (5, 3, 200),
                                     (1, 1, 256), (4, 2, 257), (5, 1, 512)])
def test_fused_decoder_stack_matches_the_layer_by_layer_path(hip, n, Lt, Lh):
    """bist_decoder_stack_fwd (one persistent launch for all decoder layers of a decode step) against the same layers run one
    launch per operation, bf16, d_model=512, h=8, on the replicated turn of a beam-search step (n hypotheses x Lt prefix tokens).
    Lh: tokens of the dialogue history -- up to 64 the attention core holds a memory's scores in registers at once, from 65 to 256 it
    walks 64-key chunks with a running maximum (decstack.hip core_unit_long, memories of up to 512 keys)."""
    from bist_amd import _lib, functional as Fn
    from bist_amd.data.batch import subsequent_mask
    from bist_amd.model.decode import _turn_for_rows
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=3, nb_venc_blocks=3, nb_cenc_blocks=3)
    V, C = 300, 256
    ob = O.det_batch(1, 8, 9, C, 20, Lh, 15, 12, V, seed=21)
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    b = to_batch(Batch, ob, torch.bfloat16)
    assert model.mutlimodal_decoder.__dict__.get("_bist_dec_state", {}).get("used") is None
    g = torch.Generator().manual_seed(n * 100 + Lt)
    trg = torch.randint(4, V, (n, Lt), generator=g).cuda()
    outs = {}
    with torch.no_grad():
        ft = model.encode(b)
        b.trg, b.trg_mask = trg[:1, :1].contiguous(), subsequent_mask(1, "cuda")
        ft = model.decode(b, ft)                                   # fills ft['_bist_reasoning'] (the first call of a turn)
        for fused in (False, True):
            Fn.FUSED_DECODE = fused
            try:
                bn, fn = _turn_for_rows(b, ft, n, {})
                bn.trg, bn.trg_mask = trg, subsequent_mask(Lt, "cuda")
                outs[fused] = model.decode(bn, dict(fn))["decoded_text"].float().cpu()
            finally:
                Fn.FUSED_DECODE = True
    assert outs[True].shape == (n, Lt, 512)
    assert model.mutlimodal_decoder.__dict__["_bist_dec_state"].get("used"), "the persistent kernel did not run"
    err = (outs[True] - outs[False]).abs().max().item()
    assert err <= 6e-2, err                                         # layer-normed outputs, bf16 storage between every operation in both
    assert torch.isfinite(outs[True]).all()
    model.mutlimodal_decoder.check_decode_errors()


def test_incremental_beam_search_equals_full_prefix_recompute(hip):
    """Beam search with one decode step at a time (the persistent decoder kernel keeps every computed row's self-attention keys /
    values in its per-layer pools; a hypothesis attends its ancestors' slots) against the reference's form (decode.py:62-66: every
    step recomputes the whole prefix of every hypothesis), bf16, d_model = 512: same n-best token lists, scores within bf16 noise
    (the keys are summed in slot order instead of (hypothesis, position) order)."""
    import bist_amd.model.decode as D
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=3, nb_venc_blocks=3, nb_cenc_blocks=3)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    outs = {}
    for seed, Lh in ((31, 30), (32, 30), (33, 150)):       # (150 history tokens: the persistent kernel's long attention core, the 150-key pointer head)
        ob = O.det_batch(1, 8, 9, C, 20, Lh, 15, 12, V, seed=seed)
        for incr in (False, True):
            D.INCREMENTAL = incr
            try:
                with torch.no_grad():
                    for _ in range(2):                 # second turn: replayed graphs, pools reused
                        hyps, best = beam_search_decode(model, to_batch(Batch, ob, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID,
                                                        beam=5, penalty=1.0, nbest=5, train_args=_args(cfg))
            finally:
                D.INCREMENTAL = True
            outs[(seed, incr)] = hyps
        a, b = outs[(seed, False)], outs[(seed, True)]
        assert len(a) == len(b) and len(a) > 0
        assert [list(map(int, x[0])) for x in a] == [list(map(int, x[0])) for x in b], seed
        assert max(abs(float(x[1]) - float(y[1])) for x, y in zip(a, b)) <= 5e-2, seed
    store = model.__dict__.get("_bist_step_graphs", {})
    assert any(k[0] == "incr" for k in store if isinstance(k, tuple)), "the incremental step graphs were not used"


@pytest.mark.parametrize("incremental", [True, False], ids=["incremental", "full_prefix"])
def test_beam_search_turns_of_different_dialogue_lengths_do_not_share_state(hip, incremental):
    """Turns of two dialogue geometries in alternation (A, B, A, B): the persistent decoder kernel's per-geometry key / value caches,
    descriptors, self-attention pools and the captured step graphs must each serve their own turn.  Every turn equals the same turn
    decoded with the layer-by-layer decoder and full-prefix steps.  full_prefix: the fused kernel on ALL prefix rows per step
    (BIST_INCREMENTAL_DECODE=0): the step graphs of both geometries share (hypotheses, prefix length) and with it the block-diagonal
    self-attention mask buffer -- replaying geometry A's graphs after geometry B's were captured must still read a live mask."""
    import bist_amd.model.decode as D
    from bist_amd import functional as Fn
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    dialogues = [O.det_batch(1, 8, 9, C, 20, 30, 15, 12, V, seed=41), O.det_batch(1, 8, 9, C, 12, 45, 10, 12, V, seed=42),
                 O.det_batch(1, 8, 9, C, 20, 30, 15, 12, V, seed=43), O.det_batch(1, 8, 9, C, 12, 45, 10, 12, V, seed=44),
                 # two more exact geometries that fall into the length classes of the first two (decode.BUCKET = "class": (32, 32, 16) and
                 # (32, 64, 16)): they REPLAY the graphs captured for other lengths, the padded tails masked
                 O.det_batch(1, 8, 9, C, 9, 17, 11, 12, V, seed=45), O.det_batch(1, 8, 9, C, 31, 64, 16, 12, V, seed=46)]

    def turn(ob):
        with torch.no_grad():
            return beam_search_decode(model, to_batch(Batch, ob, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=5,
                                      penalty=1.0, nbest=5, train_args=_args(cfg))[0]
    model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
    D.INCREMENTAL = incremental
    try:
        fast = [turn(ob) for ob in dialogues] + [turn(dialogues[0])]
        if not incremental:
            # allocations between the replays: a mask buffer freed by mistake would be recycled under the captured graphs
            junk = [torch.randn(64, 64, device="cuda") for _ in range(64)]
            again = turn(dialogues[0])
            assert [list(map(int, x[0])) for x in again] == [list(map(int, x[0])) for x in fast[0]]
            del junk
    finally:
        D.INCREMENTAL = True
    Fn.FUSED_DECODE, D.INCREMENTAL = False, False
    model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
    try:
        slow = [turn(ob) for ob in dialogues] + [turn(dialogues[0])]
    finally:
        Fn.FUSED_DECODE, D.INCREMENTAL = True, True
        model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
    for i, (a, b) in enumerate(zip(fast, slow)):
        assert [list(map(int, x[0])) for x in a] == [list(map(int, x[0])) for x in b], i
        assert max(abs(float(x[1]) - float(y[1])) for x, y in zip(a, b)) <= 5e-2, i
    assert [list(map(int, x[0])) for x in fast[0]] == [list(map(int, x[0])) for x in fast[len(dialogues)]]       # (the first dialogue once more, at the end)


@pytest.mark.parametrize("bucket", [0, 8, 16, "class"])
def test_beam_search_on_length_buckets_gives_the_same_n_best(hip, golden_dir, bucket):
    """decode.BUCKET pads the dialogue's token tensors to multiples of 8 so that dialogues of many lengths share their graphs: the
    padded positions are masked everywhere, the n-best lists and scores are those of the unpadded dialogue (fp32, reference golden)."""
    import bist_amd.model.decode as D
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g5_beam.npz"))
    meta = json.loads(str(g["beam5_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    model, _ = build_model(M, cfg, dm["V"], dm["C"])
    assert any(getattr(ob, f).shape[1] % 8 for f in ("query", "his", "cap")), "the golden dialogue would not be padded"
    old, D.BUCKET = D.BUCKET, bucket           # (0: the exact geometry; 8 is the default)
    try:
        with torch.no_grad():
            hyps, best = beam_search_decode(model, to_batch(Batch, ob), dm["maxlen"], O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=meta["beam"],
                                            penalty=1.0, nbest=5, train_args=_args(cfg))
    finally:
        D.BUCKET = old
    assert len(hyps) == int(g["beam5_n"])
    for i, (toks, score) in enumerate(hyps):
        assert [int(t) for t in toks] == g[f"beam5_hyp{i}"].tolist(), f"hyp {i}"
        assert abs(float(score) - float(g[f"beam5_score{i}"])) < 1e-3


def _counts():
    from bist_amd import _lib
    return {n: _lib.lib.bist_launch_count(getattr(_lib, n)) for n in dir(_lib) if n.startswith("K_")}


def test_persistent_decoder_steps_match_the_reference_golden(hip, golden_dir):
    """BASELINE configs[4]: the path the bench's `decode` line times -- bf16, d_model=512, L=6, the decoder layers of every decode
    step as ONE persistent launch (bist_decoder_stack_fwd), one step at a time over the per-layer self-attention pools with ancestry
    masks -- against the REFERENCE's own decode steps (tests/golden/g8_decode_steps.npz: model/decode.py:63-70 run at full size for
    five forced token sequences, every prefix length 1..12).  Step l carries the five hypotheses' NEW tokens only; hypothesis j
    attends slot 0 (<sos>, shared) and its own earlier slots.  bf16 bound: every sampled / top-16 log-prob within 8e-2 of the
    reference's fp32 value (they are O(10)); the arg-max token identical wherever the reference's own top-2 margin exceeds 2 x that."""
    import bist_amd.model.decode as D
    from bist_amd import _lib
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g8_decode_steps.npz"))
    meta = json.loads(str(g["cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    model, _ = build_model(M, cfg, dm["V"], dm["C"], torch.bfloat16)
    ob = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"] + 1)
    b = to_batch(Batch, ob, torch.bfloat16)
    seqs, sample_ids = g["seqs"], g["sample_ids"]
    n, steps = seqs.shape
    BOUND = 8e-2
    worst, decided, agree = 0.0, 0, 0

    def check(lp_row, j, l):
        nonlocal worst, decided, agree
        e = max(np.abs(lp_row[g["top_ids"][j, l]] - g["top_val"][j, l]).max(), np.abs(lp_row[sample_ids] - g["sample_val"][j, l]).max())
        worst = max(worst, float(e))
        if g["top_val"][j, l, 0] - g["top_val"][j, l, 1] > 2 * BOUND:
            decided += 1
            agree += int(lp_row.argmax()) == int(g["top_ids"][j, l, 0])

    _lib.lib.bist_launch_count_reset()
    model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
    with torch.no_grad():
        ft, lp0, _ = D._graph_first_step(model, b, O.SOS_ID, _args(cfg))
        assert ft.get("_bist_pool_ready"), "the turn's first step did not run through the persistent decoder kernel"
        for j in range(n):
            check(lp0.reshape(-1), j, 0)
        bn, fn = D._turn_for_rows(b, ft, n, {})
        for l in range(1, steps):
            slot0 = l * n
            mask = np.zeros((n, 32 if slot0 + n <= 32 else 64), dtype=np.uint8)
            for j in range(n):
                mask[j, [0] + [k * n + j for k in range(1, l)] + [slot0 + j]] = 1
            last = torch.from_numpy(seqs[:, l - 1:l].copy())
            lp = D._graph_step_incr(model, bn, fn, last, l, slot0, mask, _args(cfg))
            assert lp.shape[:2] == (n, 1)
            for j in range(n):
                check(lp[j, 0], j, l)
        model.mutlimodal_decoder.check_decode_errors()
    c = _counts()
    assert c["K_DECSTACK"] >= 2 * steps - 1, c            # warm-up + capture of the first step and of each of the 11 incremental steps
    assert worst <= BOUND, f"log-probs beyond the bf16 bound of the reference: {worst}"
    assert decided >= 30 and agree == decided, (decided, agree)


def test_bf16_persistent_decoder_beam_search_matches_the_full_size_reference_n_best(hip, golden_dir):
    """The same production path end to end: bf16 beam search (beam 5, maxlen 12; fused decoder stack + incremental steps, asserted
    through bist_launch_count) at d_model=512, L=6 on the G6 dialogue against the REFERENCE's n-best (g6_fullsize.npz T32_beam_*):
    identical token lists, scores within the bf16 bound."""
    from bist_amd import _lib
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g6_fullsize.npz"))
    meta = json.loads(str(g["T32_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    model, _ = build_model(M, cfg, dm["V"], dm["C"], torch.bfloat16)
    ob1 = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"] + 1)
    _lib.lib.bist_launch_count_reset()
    with torch.no_grad():
        for _ in range(2):                   # the second turn replays the captured graphs
            hyps, _best = beam_search_decode(model, to_batch(Batch, ob1, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID,
                                             beam=meta["beam"], penalty=1.0, nbest=5, train_args=_args(cfg))
    assert _counts()["K_DECSTACK"] >= 12, _counts()
    store = model.__dict__.get("_bist_step_graphs", {})
    assert any(k[0] == "incr" for k in store if isinstance(k, tuple)), "the incremental step graphs were not used"
    assert len(hyps) == int(g["T32_beam_n"])
    for i, (toks, score) in enumerate(hyps):
        assert [int(t) for t in toks] == g[f"T32_beam_hyp{i}"].tolist(), (i, toks)
        assert abs(float(score) - float(g[f"T32_beam_score{i}"])) <= 8e-2, (i, float(score), float(g[f"T32_beam_score{i}"]))


def test_device_side_beam_update_equals_the_host_beam_loop(hip):
    """The beam bookkeeping of decode.py:72-99 on the device (bist_beam_step: per-row top-(beam+2) of logp + lp, the replace-the-minimum
    list with the reference's visiting order, ancestry masks, step records; one copy to the host per TURN) against the same loop on the
    host (one copy and one synchronisation per STEP, numpy argsort order) -- same graphs, same log-probs: identical n-best token lists
    and bit-identical scores, beam 5 / maxlen 12 and beam 3 / maxlen 7, several dialogues, repeated turns (replayed graphs)."""
    import bist_amd.model.decode as D
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    for beam, maxlen in ((5, 12), (3, 7)):
        for seed in (51, 52, 53):
            ob = O.det_batch(1, 8, 9, C, 20, 30, 15, 12, V, seed=seed)
            outs = {}
            for dev_beam in (True, False):
                D.DEVICE_BEAM = dev_beam
                try:
                    with torch.no_grad():
                        for _ in range(2):
                            hyps, best = beam_search_decode(model, to_batch(Batch, ob, torch.bfloat16), maxlen, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID,
                                                            beam=beam, penalty=1.0, nbest=5, train_args=_args(cfg))
                finally:
                    D.DEVICE_BEAM = True
                outs[dev_beam] = (hyps, best)
            a, b = outs[True], outs[False]
            assert [list(map(int, x[0])) for x in a[0]] == [list(map(int, x[0])) for x in b[0]], (beam, seed)
            assert all(float(x[1]) == float(y[1]) for x, y in zip(a[0], b[0])), (beam, seed, a[0], b[0])
            assert float(a[1]) == float(b[1])
    store = model.__dict__.get("_bist_step_graphs", {})
    assert any(isinstance(k, tuple) and k[0] == "beam_state" for k in store), "the device-side beam path was not taken"


def test_head_local_decoder_kernel_equals_the_column_split_kernel(hip):
    """The persistent decoder kernel's head-local form (R <= 16 rows: workgroup hh owns head hh through an attention sublayer and writes a
    partial output projection; 6 grid barriers per layer; opt-in, BIST_DECSTACK_HEADLOCAL=1) against its column-split form (14 barriers
    per layer; the default: the caller passes no partial buffer) on the same decode steps: row counts 1, 5, 14, 16, with and without earlier rows in the self-attention
    pools.  Same arithmetic up to the order of the f32 sums over the heads: outputs within two bf16 ulps of the largest magnitude (the residual stream is bf16), mean difference <= 2e-3, finite."""
    from bist_amd.data.batch import subsequent_mask
    from bist_amd.model.decode import _turn_for_rows
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=3, nb_venc_blocks=3, nb_cenc_blocks=3)
    V, C = 300, 256
    ob = O.det_batch(1, 8, 9, C, 20, 30, 15, 12, V, seed=61)
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    b = to_batch(Batch, ob, torch.bfloat16)
    dec = model.mutlimodal_decoder
    with torch.no_grad():
        ft = model.encode(b)
        b.trg, b.trg_mask = torch.full((1, 1), O.SOS_ID, device="cuda"), subsequent_mask(1, "cuda")
        ft = model.decode(b, ft)
        st = dec.__dict__["_bist_dec_state"]
        for n, Lt in ((1, 1), (5, 1), (2, 7), (4, 4)):
            g = torch.Generator().manual_seed(10 * n + Lt)
            trg = torch.randint(4, V, (n, Lt), generator=g).cuda()
            outs = {}
            for head_local in (True, False):
                dec.HEAD_LOCAL = head_local
                try:
                    bn, fn = _turn_for_rows(b, ft, n, {})
                    bn.trg, bn.trg_mask = trg, subsequent_mask(Lt, "cuda")
                    outs[head_local] = model.decode(bn, dict(fn))["decoded_text"].float().cpu()
                finally:
                    dec.__dict__.pop("HEAD_LOCAL", None)
            assert torch.isfinite(outs[True]).all()
            diff = (outs[True] - outs[False]).abs()
            ulp2 = 2.0 ** -6 * outs[False].abs().max().item()            # two bf16 ulps at the largest magnitude of the rows
            assert diff.max().item() <= ulp2 and diff.mean().item() <= 2e-3, (n, Lt, diff.max().item(), diff.mean().item())
        dec.check_decode_errors()


@pytest.mark.parametrize("Ls", [(20, 61), (23, 256), (129,), (512, 300)])
def test_pointer_decode_launch_matches_its_formulas(hip, Ls):
    """bist_pointer_decode_mix_fwd (the pointer heads of a decode step in one launch: folded keys M = K W_q, c = K b_q, switch blocks
    E = enc W_sw^T) against the same formulas in torch f64 on random f32 operands: scores -> masked softmax -> switch -> log mixture
    (generator.py:84-127).  f32 arithmetic: log-probs within 2e-5 where the mixture is above 1e-30, pointer probabilities within 1e-6."""
    from bist_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, d, V = 5, 512, 301
    n, ns = len(Ls), len(Ls) + 1
    rnd = lambda *s: torch.randn(*s, generator=g)
    x, tgt, logits = rnd(rows, d), rnd(rows, d), rnd(rows, V) * 3
    wsw, bsw = rnd(ns, (n + 2) * d) * 0.05, rnd(ns)
    srcs, ref_p = [], []
    for L in Ls:
        mask = (torch.rand(L, generator=g) > 0.2).to(torch.uint8)
        srcs.append({"M": rnd(L, d) * 0.3, "c": rnd(L), "mask": mask, "E": rnd(L, ns) * 0.2, "text": torch.randint(0, V, (L,), generator=g)})
    scale = 1.0 / d ** 0.5
    X, T = x.double(), tgt.double()
    swl = X @ wsw[:, :d].double().t() + T @ wsw[:, d:2 * d].double().t() + bsw.double()
    for sj in srcs:
        sc = (X @ sj["M"].double().t() + sj["c"].double()) * scale
        sc = sc.masked_fill(sj["mask"][None] == 0, -1e9)
        p = torch.softmax(sc, -1)
        ref_p.append(p)
        swl = swl + p @ sj["E"].double()
    sw = torch.softmax(swl, -1)
    mix = sw[:, n:n + 1] * torch.softmax(logits.double(), -1)
    for j, sj in enumerate(srcs):
        mix = mix.scatter_add(1, sj["text"][None].expand(rows, -1), sw[:, j:j + 1] * ref_p[j])
    dsrc = [{k: v.cuda() for k, v in sj.items()} for sj in srcs]
    for sj, L in zip(dsrc, Ls):
        sj["p"] = torch.empty(rows, L, device="cuda")
    out = ops.pointer_decode_mix(x.cuda(), tgt.cuda(), logits.cuda(), dsrc, wsw.cuda(), bsw.cuda(), scale)
    torch.cuda.synchronize()
    for j in range(n):
        assert (dsrc[j]["p"].cpu().double() - ref_p[j]).abs().max().item() <= 1e-6
    keep = mix > 1e-30
    assert ((out.cpu().double() - mix.log()).abs()[keep]).max().item() <= 2e-5
    assert torch.isfinite(out).all()


def test_decode_step_pointer_heads_in_one_launch_equal_the_separate_launches(hip):
    """A decode step's generator with the per-turn folded keys (MultiPointerGenerator._forward_decode) against the separate launches of
    generator.py:84-127 (two projections, attention core, text vector per source, four switch products, mixture) on the same decoded
    rows, bf16 model, d=512, L=3: 1 and 5 hypothesis rows, two different dialogues in turn (the constants are rewritten in place).
    The folded form keeps q.k in f32 where the separate form rounds q, k and the text vector to bf16: log-probs within 6e-2 over the
    entries above log 1e-6 and the same arg-max token in every row."""
    from bist_amd.data.batch import subsequent_mask
    from bist_amd.model.decode import _turn_for_rows
    from bist_amd.model.generator import MultiPointerGenerator
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=3, nb_venc_blocks=3, nb_cenc_blocks=3)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    assert isinstance(model.generator, MultiPointerGenerator)
    args = _args(cfg)
    taken = 0
    for seed, (Lq, Lh) in ((61, (20, 30)), (62, (20, 30)), (63, (17, 41))):
        ob = O.det_batch(1, 8, 9, C, Lq, Lh, 15, 12, V, seed=seed)
        b = to_batch(Batch, ob, torch.bfloat16)
        with torch.no_grad():
            ft = model.encode(b)
            b.trg, b.trg_mask = torch.full((1, 1), O.SOS_ID, device="cuda"), subsequent_mask(1, "cuda")
            ft = model.decode(b, ft)                      # (leaves the per-layer reasoning results in ft)
            for n, Lt in ((1, 1), (5, 1), (1, 6)):
                bn, fn = _turn_for_rows(b, ft, n, {})
                bn.trg = torch.randint(4, V, (n, Lt), generator=torch.Generator().manual_seed(seed + n)).cuda()
                bn.trg_mask = subsequent_mask(Lt, "cuda")
                f2 = model.decode(bn, dict(fn))
                assert "_bist_turn_consts" in f2, "the persistent decoder kernel was not taken"
                outs = {}
                for fast in (True, False):
                    model.generator.DECODE_FAST = fast
                    try:
                        outs[fast] = model.generator(dict(f2), bn, args).float().cpu()
                    finally:
                        model.generator.__dict__.pop("DECODE_FAST", None)
                taken += 1
                a, r = outs[True], outs[False]
                assert a.shape == r.shape and torch.isfinite(a).all()
                keep = r > math.log(1e-6)
                assert (a - r).abs()[keep].max().item() <= 6e-2, (seed, n, Lt, (a - r).abs()[keep].max().item())
                assert torch.equal(a.argmax(-1), r.argmax(-1)), (seed, n, Lt)
    assert taken == 9


def test_beam_search_over_more_dialogue_geometries_than_the_graph_store_holds(hip):
    """decode.MAX_GEOMETRIES bounds the captured decode graphs (a test set has thousands of (query, history, caption) lengths): with room
    for two geometries, turns over three geometries in rotation (A B C A B C A) drop and re-capture the store on the way; every turn
    gives the n-best list and scores of the same turn decoded with an unbounded store, and the store never holds more than two
    first-step graphs."""
    import bist_amd.model.decode as D
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    shapes = [(20, 30, 15), (12, 45, 10), (17, 22, 30)]       # three geometries also under the default length classes: (32, 32, 16), (32, 64, 16), (32, 32, 32)
    dialogues = [O.det_batch(1, 8, 9, C, *shapes[i % 3], 12, V, seed=70 + i) for i in range(7)]

    def turns(limit):
        model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)
        old, D.MAX_GEOMETRIES = D.MAX_GEOMETRIES, limit
        out, held = [], 0
        try:
            with torch.no_grad():
                for ob in dialogues:
                    out.append(beam_search_decode(model, to_batch(Batch, ob, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=5,
                                                  penalty=1.0, nbest=5, train_args=_args(cfg))[0])
                    held = max(held, sum(1 for k in model.__dict__["_bist_step_graphs"] if isinstance(k, tuple) and k and k[0] == "first"))
        finally:
            D.MAX_GEOMETRIES = old
        return out, held
    small, held_small = turns(2)
    big, held_big = turns(48)
    assert held_small <= 2 and held_big == 3
    for i, (a, b) in enumerate(zip(small, big)):
        assert [list(map(int, x[0])) for x in a] == [list(map(int, x[0])) for x in b], i
        assert all(float(x[1]) == float(y[1]) for x, y in zip(a, b)), i
    model.__dict__.pop("_bist_step_graphs", None); model.__dict__.pop("_bist_step_graphs_key", None)


def test_a_model_that_has_decoded_can_be_saved_and_copied(hip, tmp_path):
    """After beam search the model holds captured hipGraphs, ctypes descriptors and per-turn device buffers (instance attributes named
    _bist_*).  torch.save(model) (the reference's checkpoint form, train.py:113) and copy.deepcopy leave them behind; the copy and the
    reloaded model decode the same n-best lists from their own fresh state."""
    import copy
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 256
    model, _ = build_model(M, cfg, V, C, torch.bfloat16)
    ob = O.det_batch(1, 8, 9, C, 20, 70, 15, 12, V, seed=77)

    def turn(m):
        with torch.no_grad():
            hyps, _ = beam_search_decode(m, to_batch(Batch, ob, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=5, penalty=1.0, nbest=5,
                                         train_args=_args(cfg))
        return [list(map(int, h[0])) for h in hyps]
    ref = turn(model)
    assert "_bist_step_graphs" in model.__dict__ and "_bist_dec_state" in model.mutlimodal_decoder.__dict__
    twin = copy.deepcopy(model)
    assert not any(k.startswith("_bist_") for k in twin.__dict__) and not any(k.startswith("_bist_") for k in twin.mutlimodal_decoder.__dict__)
    path = str(tmp_path / "model.pth.tar")
    torch.save(model, path)
    back = torch.load(path, weights_only=False)
    assert turn(twin) == ref and turn(back) == ref and turn(model) == ref


def test_decode_graphs_follow_the_parameters(hip):
    """The captured decode graphs bake derived operands (packed / fragment-ordered weights, descriptors).  Writing the parameters in place
    -- load_state_dict of another checkpoint into the same model -- makes them stale: the next turn (whose first-step graph is launched
    BEFORE the staleness check, for latency) must come out as the turn of a fresh model with the new weights, and so must the one after."""
    from bist_amd.model.decode import beam_search_decode
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    V, C = 300, 256
    ob = O.det_batch(1, 8, 9, C, 20, 40, 15, 12, V, seed=91)

    def turn(m):
        with torch.no_grad():
            hyps, _ = beam_search_decode(m, to_batch(Batch, ob, torch.bfloat16), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=5, penalty=1.0, nbest=5,
                                         train_args=_args(cfg))
        return [(list(map(int, h[0])), round(float(h[1]), 3)) for h in hyps]
    a, _ = build_model(M, cfg, V, C, torch.bfloat16)
    torch.manual_seed(1234)
    b = M.make_model(V, V, _args(cfg), ft_sizes=[C]).cuda().to(torch.bfloat16).eval()
    want_b = turn(b)
    first_a = turn(a)
    assert turn(a) == first_a                              # (replayed)
    a.load_state_dict(b.state_dict())
    assert turn(a) == want_b and turn(a) == want_b
    assert first_a != want_b, "the two checkpoints decode alike: the test would not see stale graphs"
