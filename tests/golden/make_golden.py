#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where /root/reference exists.  It imports the
reference's own ``model.*`` / ``data.dataset`` modules (unmodified, read-only),
loads the deterministic parameters of ``oracle.bist_oracle.det_state`` into a
``make_model(...)`` instance, runs it on the deterministic synthetic batches of
``det_batch`` and stores ONLY inputs' recipe + the reference's outputs (data,
not source) as ``*.npz``.  Weights and inputs are regenerated from their
names/seeds by the tests, so the fixtures stay small.

Capture-process-only shims (never shipped, the reference is not modified):
  * ``nltk`` is not installed and is imported-but-unused by model/decode.py:10
    -> an empty stub module;
  * generator.py:66,113 / decode.py:63-65 hard-code ``.cuda()`` -> identity on
    this CPU-only box.

usage:  python tests/golden/make_golden.py [g6|g7|g8]   (from the repo root; "g6" regenerates the full-size digests only,
        "g8" the teacher-forced decode-step log-probs of the full-size model only)
"""
import argparse
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("BIST_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

sys.modules.setdefault("nltk", types.ModuleType("nltk"))
sys.modules.setdefault("nltk.util", types.ModuleType("nltk.util"))
sys.modules["nltk.util"].ngrams = lambda *a, **k: None
torch.Tensor.cuda = lambda self, *a, **k: self

from oracle import bist_oracle as O          # noqa: E402  (det_* helpers + Cfg only)
import model.mtn as R                         # noqa: E402  the reference
import model.modules as RM                    # noqa: E402
import model.decode as RD                     # noqa: E402
from model.label_smoothing import LabelSmoothing   # noqa: E402
from model.optimize import SimpleLossCompute       # noqa: E402
from data.dataset import Batch as RBatch      # noqa: E402


def ref_args(cfg: O.Cfg):
    return argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})


def ref_batch(ob: O.OBatch) -> RBatch:
    return RBatch(ob.query, ob.his, [ob.fts.numpy()], ob.cap, ob.trg, ob.trg_y, O.PAD_ID, None, None)


def build(cfg: O.Cfg, vocab: int, C: int):
    model = R.make_model(vocab, vocab, ref_args(cfg), ft_sizes=[C])
    sd = O.det_state(cfg, vocab, C)
    ref_sd = model.state_dict()
    ref_names = {k for k in ref_sd if not k.endswith(".pe")}
    assert ref_names == set(sd), (sorted(ref_names ^ set(sd))[:10])
    for k in ref_names:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    # shared-embedding aliases must still be one tensor (mtn.py:82,90,101)
    assert model.generator.vocab_gen is model.query_embed[0].lut.weight
    model.eval()
    return model, sd


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def case_primitives(out):
    torch.manual_seed(0)
    d, h = 64, 4
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((3, 5, d)).astype(np.float32))
    ln = RM.LayerNorm(d)
    ln.a_2.data = O.det_param("prim.ln.a_2", (d,)); ln.b_2.data = O.det_param("prim.ln.b_2", (d,))
    out["prim_ln_x"] = npy(x); out["prim_ln_y"] = npy(ln(x))
    att = RM.MultiHeadedAttention(h, d, dropout=0.0).eval()
    for j in range(4):
        att.linears[j].weight.data = O.det_param(f"prim.attn.linears.{j}.weight", (d, d))
        att.linears[j].bias.data = O.det_param(f"prim.attn.linears.{j}.bias", (d,))
    q = torch.from_numpy(rs.standard_normal((3, 5, d)).astype(np.float32))
    kv = torch.from_numpy(rs.standard_normal((3, 7, d)).astype(np.float32))
    mask = torch.ones(3, 1, 7, dtype=torch.bool)
    mask[1, 0, 4:] = False          # partially masked
    mask[2, 0, :] = False           # fully masked row -> uniform softmax (modules.py:60)
    y = att(q, kv, kv, mask)
    out["prim_mha_q"] = npy(q); out["prim_mha_kv"] = npy(kv); out["prim_mha_mask"] = npy(mask)
    out["prim_mha_y"] = npy(y); out["prim_mha_p"] = npy(att.attn)
    causal = torch.tril(torch.ones(1, 5, 5, dtype=torch.bool)).expand(3, 5, 5)
    out["prim_mha_self_causal_y"] = npy(att(q, q, q, causal))
    out["prim_mha_nomask_y"] = npy(att(q, kv, kv, None))
    ff = RM.PositionwiseFeedForward(d, 4 * d, dropout=0.0).eval()
    ff.w_1.weight.data = O.det_param("prim.ff.w_1.weight", (4 * d, d)); ff.w_1.bias.data = O.det_param("prim.ff.w_1.bias", (4 * d,))
    ff.w_2.weight.data = O.det_param("prim.ff.w_2.weight", (d, 4 * d)); ff.w_2.bias.data = O.det_param("prim.ff.w_2.bias", (d,))
    out["prim_ffn_y"] = npy(ff(x))
    pe = RM.PositionalEncoding(d, 0.0).eval()
    out["prim_pe"] = npy(pe.pe[0, :9])


def case_model(tag, cfg, dims, out, with_grad=False, trace_layer=True, fully_masked=False):
    """Full forward (+ generator, losses, optional grads) through the reference."""
    vocab = dims["V"]
    model, sd = build(cfg, vocab, dims["C"])
    ob = O.det_batch(dims["B"], dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"],
                     vocab, seed=dims.get("seed", 1234), fully_masked_clip=fully_masked)
    rb = ref_batch(ob)
    assert torch.equal(rb.temporal_mask, ob.temporal_mask) and torch.equal(rb.trg_mask, ob.trg_mask)
    assert int(rb.ntokens) == int(ob.ntokens)
    traces = {}
    hooks = []
    if trace_layer and cfg.nb_venc_blocks > 0:
        vl = model.mutlimodal_decoder.v_layers[0]
        for i, sl in enumerate(vl.sublayer):
            hooks.append(sl.register_forward_hook(lambda m, a, o, i=i: traces.__setitem__(i, o.detach().clone())))
    if with_grad:
        model.vid_encoder.W.weight.requires_grad_(True)
        rb.fts.requires_grad_(True)
    ctx = torch.enable_grad() if with_grad else torch.no_grad()
    with ctx:
        ft = model.forward(rb)
        crit = LabelSmoothing(size=vocab, padding_idx=O.PAD_ID, smoothing=0.1)
        args = ref_args(cfg)
        logp = model.generator(ft, rb, args)
        lc = SimpleLossCompute(model.generator, model.ae_generator, crit, opt=None, args=args)
        # re-derive each loss term exactly as optimize.py:46-83 does
        norm, qn = rb.ntokens.float(), rb.qntokens.float()
        terms = {"out": crit(logp.contiguous().view(-1, vocab), rb.trg_y.contiguous().view(-1)) / norm}
        if cfg.auto_encoder:
            pairs = []
            if cfg.nb_cenc_blocks > 0:
                pairs.append(("cap_ae", "cap_ft"))
            if cfg.nb_venc_blocks > 0 and cfg.s2t:
                pairs.append(("temporal_ae", "temporal_ft"))
            if cfg.nb_venc_blocks > 0 and cfg.t2s:
                pairs.append(("spatial_ae", "spatial_ft"))
            for name, key in pairs:
                lp = model.ae_generator(ft, rb, args, key)
                terms[name] = crit(lp.contiguous().view(-1, vocab), rb.query.contiguous().view(-1)) / qn
        total = sum(terms.values())
        # cross-check against the reference's own aggregate (it reports un-normalised parts)
        rep = lc(ft, rb)
        assert abs(float(rep["out"]) - float(terms["out"] * norm)) < 1e-2 * max(1.0, abs(float(rep["out"])))
        if with_grad:
            total.backward()
    for hk in hooks:
        hk.remove()
    for k, v in ft.items():
        out[f"{tag}_ft_{k}"] = npy(v)
    out[f"{tag}_logp"] = npy(logp)
    for k, v in terms.items():
        out[f"{tag}_loss_{k}"] = npy(v)
    out[f"{tag}_loss_total"] = npy(total)
    for i, v in traces.items():
        out[f"{tag}_v0_sublayer{i}"] = npy(v)
    if with_grad:
        out[f"{tag}_grad_vidW"] = npy(model.vid_encoder.W.weight.grad)
        out[f"{tag}_grad_fts"] = npy(rb.fts.grad)
        vl = model.mutlimodal_decoder.v_layers[0]
        for ai in range(len(vl.attn)):
            for j in range(4):
                out[f"{tag}_grad_v0_attn{ai}_lin{j}_w"] = npy(vl.attn[ai].linears[j].weight.grad)
                out[f"{tag}_grad_v0_attn{ai}_lin{j}_b"] = npy(vl.attn[ai].linears[j].bias.grad)
        for si in range(len(vl.sublayer)):
            out[f"{tag}_grad_v0_sub{si}_a"] = npy(vl.sublayer[si].norm.a_2.grad)
        out[f"{tag}_grad_lut"] = npy(model.query_embed[0].lut.weight.grad)
        out[f"{tag}_grad_ptrW"] = npy(model.generator.pointer_gen_W.weight.grad)
    out[f"{tag}_cfg"] = np.asarray(json.dumps({"cfg": cfg.__dict__, "dims": dims, "fully_masked": fully_masked}))
    return model, sd, ob


def case_beam(tag, cfg, dims, out, beam):
    vocab = dims["V"]
    model, sd = build(cfg, vocab, dims["C"])
    ob = O.det_batch(1, dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"],
                     vocab, seed=dims.get("seed", 77))
    rb = ref_batch(ob)
    with torch.no_grad():
        hyps, best = RD.beam_search_decode(model, rb, dims["maxlen"], O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID,
                                           beam=beam, penalty=1.0, nbest=5, train_args=ref_args(cfg))
    out[f"{tag}_n"] = np.asarray(len(hyps))
    for i, (toks, score) in enumerate(hyps):
        out[f"{tag}_hyp{i}"] = np.asarray([int(t) for t in toks], dtype=np.int64)
        out[f"{tag}_score{i}"] = np.asarray(float(score))
    out[f"{tag}_best"] = np.asarray(float(best))
    out[f"{tag}_cfg"] = np.asarray(json.dumps({"cfg": cfg.__dict__, "dims": dims, "beam": beam}))


def digest(t, k=64):
    """What is kept of a full-size tensor: its element sum and absolute sum (float64), its first k values and k values at a fixed
    stride through the flattened tensor."""
    x = t.detach().double().reshape(-1)
    idx = torch.linspace(0, x.numel() - 1, k).long()
    return np.concatenate([[float(x.sum()), float(x.abs().sum())], x[:k].numpy(), x[idx].numpy()])


def case_fullsize(tag, cfg, dims, out, beam=None):
    """SURVEY 8c "G6": the reference at d_model=512, L=6 (BASELINE configs[1] / configs[3] model) regenerated from seed; only digests
    of the outputs are stored (plus the greedy argmax and, with `beam`, the n-best of a beam search on clip 0's dialogue)."""
    vocab = dims["V"]
    model, sd = build(cfg, vocab, dims["C"])
    ob = O.det_batch(dims["B"], dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"], vocab, seed=dims["seed"])
    rb = ref_batch(ob)
    with torch.no_grad():
        ft = model.forward(rb)
        logp = model.generator(ft, rb, ref_args(cfg))
    for k, v in ft.items():
        out[f"{tag}_dg_{k}"] = digest(v)
    out[f"{tag}_dg_logp"] = digest(logp)
    out[f"{tag}_argmax"] = npy(logp.argmax(-1))
    out[f"{tag}_logp_row0"] = npy(logp[0, 0])
    if beam:
        ob1 = O.det_batch(1, dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"], vocab, seed=dims["seed"] + 1)
        with torch.no_grad():
            hyps, best = RD.beam_search_decode(model, ref_batch(ob1), 12, O.SOS_ID, O.UNK_ID, O.EOS_ID, O.PAD_ID, beam=beam, penalty=1.0,
                                               nbest=5, train_args=ref_args(cfg))
        out[f"{tag}_beam_n"] = np.asarray(len(hyps))
        for i, (toks, score) in enumerate(hyps):
            out[f"{tag}_beam_hyp{i}"] = np.asarray([int(t) for t in toks], dtype=np.int64)
            out[f"{tag}_beam_score{i}"] = np.asarray(float(score))
    out[f"{tag}_cfg"] = np.asarray(json.dumps({"cfg": cfg.__dict__, "dims": dims, "beam": beam}))


SMALL = dict(B=2, T=6, S=9, C=48, Lq=7, Lh=11, Lc=8, Lt=6, V=60)
FULL = dict(B=4, S=49, C=2048, Lq=20, Lh=60, Lc=25, Lt=20, V=3000, seed=4242)
MID = dict(B=3, T=8, S=49, C=64, Lq=20, Lh=24, Lc=12, Lt=10, V=120)


def main_g7():
    """A whole-module pickle written by the REFERENCE (train.py:161 ``torch.save(model, path)``) for a tiny model, with its outputs:
    generate.py:93 unpickles such a file by ``model.*`` class paths, which INTEGRATION.md section 2 maps onto this build."""
    cfg = O.Cfg(d_model=16, att_h=2, nb_blocks=1, nb_venc_blocks=1, nb_cenc_blocks=1)
    dims = dict(B=2, T=4, S=9, C=8, Lq=5, Lh=6, Lc=4, Lt=5, V=30, seed=5)
    model, sd = build(cfg, dims["V"], dims["C"])
    torch.save(model, os.path.join(HERE, "g7_reference_module.pth.tar"))
    ob = O.det_batch(dims["B"], dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"], dims["V"], seed=dims["seed"])
    rb = ref_batch(ob)
    out = {}
    with torch.no_grad():
        ft = model.forward(rb)
        out["logp"] = npy(model.generator(ft, rb, ref_args(cfg)))
    for k, v in ft.items():
        out[f"ft_{k}"] = npy(v)
    out["cfg"] = np.asarray(json.dumps({"cfg": cfg.__dict__, "dims": dims}))
    np.savez_compressed(os.path.join(HERE, "g7_reference_module.npz"), **out)


def main_g6():
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=6, nb_venc_blocks=6, nb_cenc_blocks=6)
    out = {}
    case_fullsize("T32", cfg, dict(FULL, T=32), out, beam=5)
    case_fullsize("T128", cfg, dict(FULL, T=128), out)
    np.savez_compressed(os.path.join(HERE, "g6_fullsize.npz"), **out)


G8_ROWS, G8_STEPS, G8_TOPK, G8_SAMPLES = 5, 12, 16, 64


def g8_sequences(vocab):
    """Five forced responses of twelve tokens (ids 4 .. V-1: no <pad>/<sos>/<eos>/<unk>), one per hypothesis row."""
    rs = np.random.RandomState(8008)
    return rs.randint(4, vocab, size=(G8_ROWS, G8_STEPS)).astype(np.int64)


def main_g8():
    """The reference's decode STEP at d_model=512, L=6 (BASELINE configs[4]'s model): for five forced token sequences and every prefix
    length 1..12 (decode.py:63-70: trg = <sos> + prefix, causal mask, model.decode, last position, generator), the log-prob row's
    16 largest entries (ids + values) and its values at 64 fixed ids.  Pins the build's persistent decoder kernel (one decode step at
    a time, self-attention pools, ancestry masks) to the reference at every position, not only through a short n-best list."""
    from data.data_utils import subsequent_mask
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=6, nb_venc_blocks=6, nb_cenc_blocks=6)
    dims = dict(FULL, T=32)
    vocab = dims["V"]
    model, sd = build(cfg, vocab, dims["C"])
    ob1 = O.det_batch(1, dims["T"], dims["S"], dims["C"], dims["Lq"], dims["Lh"], dims["Lc"], dims["Lt"], vocab, seed=dims["seed"] + 1)
    rb = ref_batch(ob1)
    seqs = g8_sequences(vocab)
    sample_ids = np.linspace(0, vocab - 1, G8_SAMPLES).astype(np.int64)
    top_ids = np.zeros((G8_ROWS, G8_STEPS, G8_TOPK), dtype=np.int64)
    top_val = np.zeros((G8_ROWS, G8_STEPS, G8_TOPK), dtype=np.float32)
    samp = np.zeros((G8_ROWS, G8_STEPS, G8_SAMPLES), dtype=np.float32)
    args = ref_args(cfg)
    with torch.no_grad():
        ft = model.encode(rb)
        for j in range(G8_ROWS):
            for l in range(G8_STEPS):
                st = torch.tensor([[O.SOS_ID] + [int(t) for t in seqs[j, :l]]], dtype=torch.long)
                rb.trg = st
                rb.trg_mask = subsequent_mask(st.size(1)).long()
                rb.trg_mean_mask = torch.ones(st.shape).long()
                ft = model.decode(rb, ft)
                ft["decoded_text"] = ft["decoded_text"][:, -1].unsqueeze(1)
                ft["encoded_tgt"] = ft["encoded_tgt"][:, -1].unsqueeze(1)
                lp = model.generator(ft, rb, args).reshape(-1).numpy()
                order = np.argsort(lp)[::-1][:G8_TOPK]
                top_ids[j, l], top_val[j, l], samp[j, l] = order, lp[order], lp[sample_ids]
    out = {"seqs": seqs, "top_ids": top_ids, "top_val": top_val, "sample_ids": sample_ids, "sample_val": samp,
           "cfg": np.asarray(json.dumps({"cfg": cfg.__dict__, "dims": dims}))}
    np.savez_compressed(os.path.join(HERE, "g8_decode_steps.npz"), **out)


def main():
    os.makedirs(HERE, exist_ok=True)
    if sys.argv[1:] == ["g6"]:                 # the full-size digests alone (the other fixtures are untouched)
        main_g6()
        print("g6_fullsize.npz", os.path.getsize(os.path.join(HERE, "g6_fullsize.npz")) // 1024, "KiB")
        return
    if sys.argv[1:] == ["g8"]:
        main_g8()
        print("g8_decode_steps.npz", os.path.getsize(os.path.join(HERE, "g8_decode_steps.npz")) // 1024, "KiB")
        return
    if sys.argv[1:] == ["g7"]:
        main_g7()
        print("g7_reference_module.pth.tar", os.path.getsize(os.path.join(HERE, "g7_reference_module.pth.tar")) // 1024, "KiB")
        return
    out = {}
    case_primitives(out)
    np.savez_compressed(os.path.join(HERE, "g1_primitives.npz"), **out)

    out = {}
    cfg = O.Cfg(d_model=64, att_h=4, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    case_model("both", cfg, SMALL, out, with_grad=True)
    case_model("mid", O.Cfg(d_model=64, att_h=8, nb_blocks=1, nb_venc_blocks=1, nb_cenc_blocks=1), MID, out)
    case_model("t2s", O.Cfg(d_model=64, att_h=4, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2, s2t=0), SMALL, out)
    case_model("s2t", O.Cfg(d_model=64, att_h=4, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2, t2s=0), SMALL, out)
    case_model("masked", cfg, SMALL, out, fully_masked=True)
    np.savez_compressed(os.path.join(HERE, "g3_model.npz"), **out)

    out = {}
    bd = dict(SMALL, maxlen=8, seed=77)
    case_beam("beam5", cfg, bd, out, beam=5)
    case_beam("beam1", cfg, bd, out, beam=1)
    np.savez_compressed(os.path.join(HERE, "g5_beam.npz"), **out)
    main_g6()
    main_g7()
    main_g8()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
