"""Pin the CPU oracle against the vectors captured from the reference itself.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which ran the
reference's own model code (model/*.py) on deterministic weights and inputs.  Here the oracle
regenerates the same weights/inputs from their names/seeds and must reproduce the reference's
outputs.  fp32 on CPU on both sides -> tolerance 2e-5 absolute on O(1) activations.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import bist_oracle as O

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _close(a, b, tol=TOL, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - b).max() if a.size else 0.0
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err <= tol * max(1.0, float(np.abs(b).max())), f"{what}: max err {err}"


def _prim_sd(d):
    sd = {"ln.a_2": O.det_param("prim.ln.a_2", (d,)), "ln.b_2": O.det_param("prim.ln.b_2", (d,))}
    for j in range(4):
        sd[f"attn.linears.{j}.weight"] = O.det_param(f"prim.attn.linears.{j}.weight", (d, d))
        sd[f"attn.linears.{j}.bias"] = O.det_param(f"prim.attn.linears.{j}.bias", (d,))
    sd["ff.w_1.weight"] = O.det_param("prim.ff.w_1.weight", (4 * d, d)); sd["ff.w_1.bias"] = O.det_param("prim.ff.w_1.bias", (4 * d,))
    sd["ff.w_2.weight"] = O.det_param("prim.ff.w_2.weight", (d, 4 * d)); sd["ff.w_2.bias"] = O.det_param("prim.ff.w_2.bias", (d,))
    return sd


def test_primitives(golden_dir):
    g = _load(golden_dir, "g1_primitives.npz")
    d, h = 64, 4
    sd = _prim_sd(d)
    x = torch.from_numpy(g["prim_ln_x"])
    _close(O.layer_norm(x, sd["ln.a_2"], sd["ln.b_2"]), g["prim_ln_y"], what="LayerNorm")
    q, kv = torch.from_numpy(g["prim_mha_q"]), torch.from_numpy(g["prim_mha_kv"])
    mask = torch.from_numpy(g["prim_mha_mask"])
    y, p = O.mha(sd, "attn", h, q, kv, kv, mask)
    _close(y, g["prim_mha_y"], what="mha out"); _close(p, g["prim_mha_p"], what="mha p_attn")
    # the fully masked row is uniform, not NaN (modules.py:60 uses -1e9)
    assert np.allclose(p[2].numpy(), 1.0 / kv.shape[1], atol=1e-7)
    causal = torch.tril(torch.ones(1, 5, 5, dtype=torch.bool)).expand(3, 5, 5)
    _close(O.mha(sd, "attn", h, q, q, q, causal)[0], g["prim_mha_self_causal_y"], what="mha causal")
    _close(O.mha(sd, "attn", h, q, kv, kv, None)[0], g["prim_mha_nomask_y"], what="mha nomask")
    _close(O.ffn(sd, "ff", x), g["prim_ffn_y"], what="ffn")
    _close(O.pos_encoding(9, d), g["prim_pe"], what="pos enc")


def _setup(g, tag):
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg = O.Cfg(**meta["cfg"])
    dm = meta["dims"]
    sd = O.det_state(cfg, dm["V"], dm["C"])
    b = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"],
                    seed=dm.get("seed", 1234), fully_masked_clip=meta.get("fully_masked", False))
    return cfg, dm, sd, b


@pytest.mark.parametrize("tag", ["both", "mid", "t2s", "s2t", "masked"])
def test_model_forward(golden_dir, tag):
    g = _load(golden_dir, "g3_model.npz")
    cfg, dm, sd, b = _setup(g, tag)
    trace = {}
    with torch.no_grad():
        ft = O.mtn_forward(sd, cfg, b, trace)
        losses = O.loss_compute(sd, cfg, ft, b, dm["V"])
    for k in [k for k in g.files if k.startswith(f"{tag}_ft_")]:
        _close(ft[k[len(tag) + 4:]], g[k], what=k)
    _close(losses["logp"], g[f"{tag}_logp"], tol=1e-4, what="logp")
    for k in [k for k in g.files if k.startswith(f"{tag}_loss_")]:
        _close(losses[k[len(tag) + 6:]], g[k], tol=1e-4, what=k)
    # per-stage outputs of v_layers[0] (sublayer order A0,A1,A2,F0,A3,A4,A5,F1 -- encoder.py:173)
    tr = trace["v0"]
    names = []
    if cfg.t2s:
        names += ["t2s_self", "t2s_stage1", "t2s_stage2", None]
    if cfg.s2t:
        names += ["s2t_self", "s2t_stage1", "s2t_stage2", None]
    for i, n in enumerate(names):
        key = f"{tag}_v0_sublayer{i}"
        if n is None or key not in g.files:
            continue
        ref = g[key]
        got = tr[n]
        _close(got.reshape(ref.shape), ref, what=f"{tag} sublayer{i} ({n})")


def test_model_gradients(golden_dir):
    g = _load(golden_dir, "g3_model.npz")
    cfg, dm, sd, b = _setup(g, "both")
    # independent leaves, re-aliased, so shared-embedding grads accumulate like the reference's
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()
            if k not in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj")}
    for alias in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj"):
        leaf[alias] = leaf["query_embed.0.lut.weight"]
    b.fts.requires_grad_(True)
    ft = O.mtn_forward(leaf, cfg, b)
    O.loss_compute(leaf, cfg, ft, b, dm["V"])["total"].backward()
    gt = 2e-4
    _close(leaf["vid_encoder.W.weight"].grad, g["both_grad_vidW"], tol=gt, what="dW vid")
    _close(b.fts.grad, g["both_grad_fts"], tol=gt, what="d fts")
    _close(leaf["query_embed.0.lut.weight"].grad, g["both_grad_lut"], tol=gt, what="d lut")
    _close(leaf["generator.pointer_gen_W.weight"].grad, g["both_grad_ptrW"], tol=gt, what="d ptrW")
    for ai in range(6):
        for j in range(4):
            w = leaf[f"mutlimodal_decoder.v_layers.0.attn.{ai}.linears.{j}.weight"].grad
            bias = leaf[f"mutlimodal_decoder.v_layers.0.attn.{ai}.linears.{j}.bias"].grad
            _close(w, g[f"both_grad_v0_attn{ai}_lin{j}_w"], tol=gt, what=f"attn{ai}.lin{j}.w")
            _close(bias, g[f"both_grad_v0_attn{ai}_lin{j}_b"], tol=gt, what=f"attn{ai}.lin{j}.b")
    # the key bias has exactly zero gradient in exact arithmetic (softmax shift invariance)
    assert np.abs(g["both_grad_v0_attn1_lin1_b"]).max() < 1e-5


@pytest.mark.parametrize("tag", ["beam5", "beam1"])
def test_beam_search(golden_dir, tag):
    g = _load(golden_dir, "g5_beam.npz")
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    sd = O.det_state(cfg, dm["V"], dm["C"])
    b = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    with torch.no_grad():
        hyps, best = O.beam_search(sd, cfg, b, dm["maxlen"], beam=meta["beam"], penalty=1.0, nbest=5)
    assert len(hyps) == int(g[f"{tag}_n"])
    for i, (toks, score) in enumerate(hyps):
        assert [int(t) for t in toks] == g[f"{tag}_hyp{i}"].tolist(), f"hyp {i}"
        assert abs(float(score) - float(g[f"{tag}_score{i}"])) < 1e-3
    assert abs(float(best) - float(g[f"{tag}_best"])) < 1e-3


def _digest(t, k=64):
    """tests/golden/make_golden.py::digest -- what the full-size fixtures keep of a tensor."""
    x = t.detach().double().reshape(-1)
    idx = torch.linspace(0, x.numel() - 1, k).long()
    return np.concatenate([[float(x.sum()), float(x.abs().sum())], x[:k].numpy(), x[idx].numpy()])


def check_digest(got, ref, what, tol):
    """sum / absolute sum relative to the absolute sum, the 128 sampled values absolutely."""
    assert abs(got[0] - ref[0]) <= tol * max(1.0, ref[1]) * 1e-2 + tol, (what, "sum", got[0], ref[0])
    assert abs(got[1] - ref[1]) <= tol * max(1.0, ref[1]) * 1e-2 + tol, (what, "abs sum", got[1], ref[1])
    err = np.abs(got[2:] - ref[2:]).max()
    assert err <= tol, (what, "samples", err)


@pytest.mark.parametrize("tag", ["T32", "T128"])
def test_oracle_matches_reference_at_full_size(golden_dir, tag):
    """d_model=512, L=6, T=32 / T=128 (BASELINE configs[1] / configs[3] model), regenerated from seed; the
    reference's outputs are pinned by digests (tests/golden/g6_fullsize.npz)."""
    g = np.load(os.path.join(golden_dir, "g6_fullsize.npz"))
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    sd = O.det_state(cfg, dm["V"], dm["C"])
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    with torch.no_grad():
        ft = O.mtn_forward(sd, cfg, ob)
        logp = O.multi_pointer_generator(sd, cfg, ft, ob)
    for k in ft:
        check_digest(_digest(ft[k]), g[f"{tag}_dg_{k}"], f"{tag} {k}", 2e-4)
    check_digest(_digest(logp), g[f"{tag}_dg_logp"], f"{tag} logp", 2e-4)
    assert np.array_equal(logp.argmax(-1).numpy(), g[f"{tag}_argmax"])
    assert np.abs(logp[0, 0].numpy() - g[f"{tag}_logp_row0"]).max() <= 2e-4


def test_oracle_decode_steps_match_reference_at_full_size(golden_dir):
    """tests/golden/g8_decode_steps.npz: the REFERENCE's decode step (decode.py:63-70) at d_model=512, L=6 for five forced token
    sequences and every prefix length 1..12 -- the 16 largest log-probs (ids + values) and 64 sampled entries per step.  The oracle
    is checked on a subset (two rows, three prefix lengths: each oracle decode re-runs the whole reasoning, like the reference)."""
    g = np.load(os.path.join(golden_dir, "g8_decode_steps.npz"))
    meta = json.loads(str(g["cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    sd = O.det_state(cfg, dm["V"], dm["C"])
    b = O.det_batch(1, dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"] + 1)
    seqs, sample_ids = g["seqs"], g["sample_ids"]
    with torch.no_grad():
        ft = O.mtn_encode(sd, cfg, b)
        for j, l in ((0, 0), (0, 5), (0, 11), (3, 1), (3, 7), (3, 11)):
            st = torch.tensor([[O.SOS_ID] + [int(t) for t in seqs[j, :l]]], dtype=torch.long)
            b.trg, b.trg_mask = st, O.subsequent_mask(st.size(1))
            ft = O.mtn_decode(sd, cfg, b, ft)
            step = dict(ft)
            step["decoded_text"], step["encoded_tgt"] = ft["decoded_text"][:, -1:], ft["encoded_tgt"][:, -1:]
            lp = O.multi_pointer_generator(sd, cfg, step, b).reshape(-1).numpy()
            assert np.abs(lp[g["top_ids"][j, l]] - g["top_val"][j, l]).max() <= 2e-4, (j, l)
            assert np.abs(lp[sample_ids] - g["sample_val"][j, l]).max() <= 2e-4, (j, l)
            assert int(lp.argmax()) == int(g["top_ids"][j, l, 0]), (j, l)
