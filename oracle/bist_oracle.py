"""CPU oracle for the BiST hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) *restatement* of the reference
algorithm for the bi-directional spatio-temporal attention path of
salesforce/BiST, written from the reference's behaviour and citing the
reference file:line each function follows.  It is the checker the parity tests
compare the HIP path against; it is never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``bist_amd/`` imports it.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so this oracle is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference``) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks
every function here against those vectors.

Style: functional.  Parameters live in a flat ``dict[str, Tensor]`` whose keys
are exactly the reference model's ``state_dict()`` names, so a
reference checkpoint's ``state_dict`` can be passed in directly.  Everything is
differentiable torch code, so ``torch.autograd`` on the oracle provides the
reference gradients as well.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

PAD_ID = 1   # '<blank>'  (data/data_handler.py:23)
UNK_ID = 0   # '<unk>'
SOS_ID = 2
EOS_ID = 3


# ----------------------------------------------------------------------------
# configuration (the subset of the reference's argparse namespace the path reads)
# ----------------------------------------------------------------------------
@dataclass
class Cfg:
    """Mirror of the flags read at forward time (configs/train_configs.py:27-46)."""
    d_model: int = 512
    att_h: int = 8
    nb_blocks: int = 6
    nb_venc_blocks: int = 6
    nb_cenc_blocks: int = 6
    nb_aenc_blocks: int = 0
    t2s: int = 1
    s2t: int = 1
    ptr_gen: int = 1
    ptr_ft: str = "query,cap"
    mask_unk: int = 1
    auto_encoder: int = 1
    include_caption: str = "summary"
    enc_st_combine: str = "none"
    dec_st_combine: str = "seq"
    enc_vc_combine: str = "dyn"
    dropout: float = 0.0

    def check_scope(self) -> None:
        # SURVEY.md 8(a) row VL / appendix: only these settings are runnable for L>1
        assert self.enc_st_combine == "none"
        assert self.nb_aenc_blocks == 0
        assert self.ptr_gen == 1
        assert self.t2s or self.s2t


@dataclass
class OBatch:
    """The Batch field contract the path consumes (data/dataset.py:59-99)."""
    query: Tensor            # [B, Lq] int64
    his: Tensor              # [B, Lh]
    cap: Tensor              # [B, Lc]
    trg: Tensor              # [B, Lt]
    trg_y: Optional[Tensor]  # [B, Lt]
    fts: Tensor              # [B, T, S, C] float32
    query_mask: Tensor = field(init=False)
    his_mask: Tensor = field(init=False)
    cap_mask: Tensor = field(init=False)
    temporal_mask: Tensor = field(init=False)
    trg_mask: Tensor = field(init=False)

    def __post_init__(self):
        pad = PAD_ID
        self.query_mask = (self.query != pad).unsqueeze(-2)          # dataset.py:66
        self.his_mask = (self.his != pad).unsqueeze(-2)              # dataset.py:67
        self.cap_mask = (self.cap != pad).unsqueeze(-2)              # dataset.py:92
        # dataset.py:79  a temporal step is "present" iff its features are not all-zero
        self.temporal_mask = (self.fts.sum(2).sum(-1) != 0).unsqueeze(-2)
        self.trg_mask = make_std_mask(self.trg, pad)                 # dataset.py:96
        if self.trg_y is not None:
            self.ntokens = (self.trg_y != pad).sum()                 # dataset.py:98
        self.qntokens = (self.query != pad).sum()                    # dataset.py:99


def subsequent_mask(size: int) -> Tensor:
    """data/data_utils.py:14-18 -- True on and below the diagonal, shape [1,size,size]."""
    return torch.tril(torch.ones(1, size, size, dtype=torch.bool))


def make_std_mask(trg: Tensor, pad: int) -> Tensor:
    """data/dataset.py:101-105 -- pad mask AND causal mask -> [B, Lt, Lt]."""
    return (trg != pad).unsqueeze(-2) & subsequent_mask(trg.size(-1))


# ----------------------------------------------------------------------------
# primitives (model/modules.py)
# ----------------------------------------------------------------------------
# Training-mode dropout (tests only): the reference drops at four kinds of site -- the sum of position encoding and embedding
# (modules.py:144, kind "pe"), the attention probabilities after the softmax (modules.py:62-63, "attn"), the feed-forward block's
# hidden layer (modules.py:113, "ffn") and every sublayer's output before the residual (modules.py:44, "sub").  With DROP_HOOK set
# -- callable(kind, site name, tensor) -> tensor with that site's mask and 1/(1-p) applied -- the oracle passes the tensor of every
# such site through it (site name: the module's state_dict path; None for "pe", whose sites come in the order query, caption,
# history, target).  tests/test_dropout_parity_gpu.py feeds it the masks the HIP kernels drew.  None (default): evaluation mode.
DROP_HOOK = None


def _drop(kind: str, name: Optional[str], x: Tensor) -> Tensor:
    return x if DROP_HOOK is None else DROP_HOOK(kind, name, x)


def layer_norm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """modules.py:28-31.  NOT F.layer_norm: unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).sum(-1, keepdim=True) / (x.size(-1) - 1)
    return a * (x - mean) / (var.sqrt() + eps) + b


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return layer_norm(x, sd[p + ".a_2"], sd[p + ".b_2"])


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return x @ sd[p + ".weight"].t() + sd[p + ".bias"]


def mha(sd: SD, p: str, h: int, query: Tensor, key: Tensor, value: Tensor,
        mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """modules.py:81-100 + attention() modules.py:54-64 (eval mode, dropout off).

    query [N,Lq,d], key/value [N,Lk,d], mask broadcastable to [N,Lq,Lk] or None.
    Returns (output [N,Lq,d], p_attn [N,h,Lq,Lk]); masked scores become -1e9
    (not -inf), so a fully masked row is a uniform distribution.
    """
    n, lq, d = query.shape
    dk = d // h

    def split(x):
        return x.reshape(n, -1, h, dk).transpose(1, 2)

    q = split(_lin(sd, p + ".linears.0", query))
    k = split(_lin(sd, p + ".linears.1", key))
    v = split(_lin(sd, p + ".linears.2", value))
    scores = q @ k.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p_attn = _drop("attn", p, torch.softmax(scores, dim=-1))                   # modules.py:62-63
    ctx = (p_attn @ v).transpose(1, 2).reshape(n, lq, d)
    return _lin(sd, p + ".linears.3", ctx), p_attn


def ffn(sd: SD, p: str, x: Tensor) -> Tensor:
    """modules.py:112-113."""
    return _lin(sd, p + ".w_2", _drop("ffn", p, torch.relu(_lin(sd, p + ".w_1", x))))


def pos_encoding(length: int, d: int) -> Tensor:
    """modules.py:131-138 -- sinusoidal table rows [0,length)."""
    pos = torch.arange(0.0, length).unsqueeze(1)
    div = torch.exp(torch.arange(0.0, d, 2) * -(math.log(10000.0) / d))
    pe = torch.zeros(length, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def embed(sd: SD, ids: Tensor, d: int) -> Tensor:
    """modules.py:121-123 + 141-144: lut[ids]*sqrt(d) + PE (dropout off)."""
    lut = sd["query_embed.0.lut.weight"]        # == tgt_embed.0.lut.weight (mtn.py:82)
    return _drop("pe", None, lut[ids] * math.sqrt(d) + pos_encoding(ids.size(1), d).to(lut.dtype))


# ----------------------------------------------------------------------------
# encoders (model/encoder.py)
# ----------------------------------------------------------------------------
def encode_text(sd: SD, cfg: Cfg, b: OBatch) -> Dict[str, Tensor]:
    """mtn.py:42-47 + Encoder.forward encoder.py:19-41: one LayerNorm each, order query,cap,his."""
    d = cfg.d_model
    return {
        "encoded_query": _ln(sd, "text_encoder.norm.0", embed(sd, b.query, d)),
        "encoded_cap": _ln(sd, "text_encoder.norm.1", embed(sd, b.cap, d)),
        "encoded_his": _ln(sd, "text_encoder.norm.2", embed(sd, b.his, d)),
    }


def vid_input_proj(sd: SD, fts: Tensor) -> Tensor:
    """P0: VidEncoder8.forward encoder.py:72-81: LN(ReLU(W fts + b)), [B,T,S,C]->[B,T,S,d]."""
    return _ln(sd, "vid_encoder.in_norm", torch.relu(_lin(sd, "vid_encoder.W", fts)))


def st_stage1(sd: SD, lp: str, h: int, a: int, s: int, x: Tensor, kv: Tensor,
              mask: Optional[Tensor]) -> Tensor:
    """A1 / A4 (encoder.py:110-123 / 142-150).

    x  [B,Lq,d]  query stream; kv [B,G,K,d] video features grouped so that group
    g attends over its K keys (t2s: G=S,K=T; s2t: G=T,K=S); mask [B,1,K] or None.
    Every group gets x + MHA(LN(x), kv_g, kv_g): the residual is the expanded query.
    Returns [B,G,Lq,d].
    """
    B, G, K, d = kv.shape
    Lq = x.size(1)
    xe = x.unsqueeze(1).expand(B, G, Lq, d).reshape(B * G, Lq, d)
    kvf = kv.reshape(B * G, K, d)
    m = None
    if mask is not None:
        m = mask.unsqueeze(1).expand(B, G, 1, K).reshape(B * G, 1, K)
    out, _ = mha(sd, f"{lp}.attn.{a}", h, _ln(sd, f"{lp}.sublayer.{s}.norm", xe), kvf, kvf, m)
    return (xe + _drop("sub", f"{lp}.sublayer.{s}", out)).reshape(B, G, Lq, d)


def st_stage2(sd: SD, lp: str, h: int, a: int, s: int, x: Tensor, y: Tensor,
              mask: Optional[Tensor]) -> Tensor:
    """A2 / A5 (encoder.py:125-134 / 152-165).

    x [B,Lq,d]; y [B,G,Lq,d] = stage-1 output.  Query position i attends, alone,
    over the G stage-1 outputs of its own position.  mask [B,1,G] or None.
    Returns [B,Lq,d].
    """
    B, G, Lq, d = y.shape
    keys = y.permute(0, 2, 1, 3).reshape(B * Lq, G, d)
    q = x.reshape(B * Lq, 1, d)
    m = None
    if mask is not None:
        m = mask.unsqueeze(1).expand(B, Lq, 1, G).reshape(B * Lq, 1, G)
    out, _ = mha(sd, f"{lp}.attn.{a}", h, _ln(sd, f"{lp}.sublayer.{s}.norm", q), keys, keys, m)
    return (q + _drop("sub", f"{lp}.sublayer.{s}", out)).reshape(B, Lq, d)


def vid_layer(sd: SD, cfg: Cfg, lp: str, in_ft: Dict[str, Tensor], vft: Tensor,
              b: OBatch, trace: Optional[dict] = None) -> Dict[str, Tensor]:
    """VidEncoderLayer4.forward encoder.py:172-201 (enc_st_combine == 'none').

    Index order (encoder.py:173 run-time counters): with both directions on,
    attn 0..5 = A0,A1,A2,A3,A4,A5; sublayer 0..7 = A0,A1,A2,F0,A3,A4,A5,F1;
    ff 0,1.  With one direction only the indices start again from 0.
    """
    h = cfg.att_h
    ai = si = fi = 0

    def self_attn(x):
        nonlocal ai, si
        n = _ln(sd, f"{lp}.sublayer.{si}.norm", x)
        o, _ = mha(sd, f"{lp}.attn.{ai}", h, n, n, n, b.query_mask)
        o = _drop("sub", f"{lp}.sublayer.{si}", o)
        ai += 1
        si += 1
        return x + o

    def ff_block(x):
        nonlocal si, fi
        o = _drop("sub", f"{lp}.sublayer.{si}", ffn(sd, f"{lp}.ff.{fi}", _ln(sd, f"{lp}.sublayer.{si}.norm", x)))
        si += 1
        fi += 1
        return x + o

    out = dict(in_ft)
    if cfg.t2s:
        x = self_attn(in_ft["t2s"])                                   # A0  encoder.py:176
        y = st_stage1(sd, lp, h, ai, si, x, vft.permute(0, 2, 1, 3), b.temporal_mask)  # A1
        ai += 1; si += 1
        z = st_stage2(sd, lp, h, ai, si, x, y, None)                  # A2
        ai += 1; si += 1
        if trace is not None:
            trace.update(t2s_self=x, t2s_stage1=y, t2s_stage2=z)
        out["t2s"] = ff_block(z)                                      # F0  encoder.py:135
    if cfg.s2t:
        x = self_attn(in_ft["s2t"])                                   # A3  encoder.py:184
        y = st_stage1(sd, lp, h, ai, si, x, vft, None)                # A4
        ai += 1; si += 1
        z = st_stage2(sd, lp, h, ai, si, x, y, b.temporal_mask)       # A5
        ai += 1; si += 1
        if trace is not None:
            trace.update(s2t_self=x, s2t_stage1=y, s2t_stage2=z)
        out["s2t"] = ff_block(z)                                      # F1  encoder.py:166
    return out


def cap_layer(sd: SD, cfg: Cfg, lp: str, c: Tensor, enc_cap: Tensor, b: OBatch) -> Tensor:
    """CapEncoderLayer.forward encoder.py:211-218."""
    h = cfg.att_h
    n = _ln(sd, f"{lp}.sublayer.0.norm", c)
    c = c + _drop("sub", f"{lp}.sublayer.0", mha(sd, f"{lp}.attn.0", h, n, n, n, b.query_mask)[0])
    c = c + _drop("sub", f"{lp}.sublayer.1", mha(sd, f"{lp}.attn.1", h, _ln(sd, f"{lp}.sublayer.1.norm", c), enc_cap, enc_cap, b.cap_mask)[0])
    return c + _drop("sub", f"{lp}.sublayer.2", ffn(sd, f"{lp}.ff", _ln(sd, f"{lp}.sublayer.2.norm", c)))


# ----------------------------------------------------------------------------
# decoder (model/decoder.py)
# ----------------------------------------------------------------------------
def dec_layer(sd: SD, cfg: Cfg, lp: str, b: OBatch, ft: Dict[str, Tensor], x: Tensor) -> Tensor:
    """MultimodalDecoderLayer12.forward decoder.py:20-60."""
    h = cfg.att_h

    def cross(i, x, mem, mask):
        return x + _drop("sub", f"{lp}.sublayer.{i}", mha(sd, f"{lp}.attn.{i}", h, _ln(sd, f"{lp}.sublayer.{i}.norm", x), mem, mem, mask)[0])

    n = _ln(sd, f"{lp}.sublayer.0.norm", x)
    x = x + _drop("sub", f"{lp}.sublayer.0", mha(sd, f"{lp}.attn.0", h, n, n, n, b.trg_mask)[0])            # decoder.py:21
    x = cross(1, x, ft["encoded_his"], b.his_mask)                         # :22
    x = cross(2, x, ft["encoded_query"], b.query_mask)                     # :23
    cnt = 3
    if cfg.nb_venc_blocks > 0 and cfg.nb_cenc_blocks > 0 and cfg.enc_vc_combine != "none":
        x = cross(cnt, x, ft["encoded_ft"], b.query_mask); cnt += 1       # :27-29
    else:
        if cfg.include_caption != "none":                                   # :31-36
            if cfg.nb_cenc_blocks > 0:
                x = cross(cnt, x, ft["cap_ft"], b.query_mask)
            else:
                x = cross(cnt, x, ft["encoded_cap"], b.cap_mask)
            cnt += 1
        if cfg.nb_venc_blocks > 0:                                          # :37-51
            if cfg.dec_st_combine == "seq":
                if cfg.s2t:
                    x = cross(cnt, x, ft["temporal_ft"], b.query_mask); cnt += 1
                if cfg.t2s:
                    x = cross(cnt, x, ft["spatial_ft"], b.query_mask); cnt += 1
            else:
                tx = cross(cnt, x, ft["temporal_ft"], b.query_mask); cnt += 1
                sx = cross(cnt, x, ft["spatial_ft"], b.query_mask); cnt += 1
                x = tx + sx
    return x + _drop("sub", f"{lp}.sublayer.{cnt}", ffn(sd, f"{lp}.ff", _ln(sd, f"{lp}.sublayer.{cnt}.norm", x)))   # :58


def fuse_modalities(sd: SD, cfg: Cfg, ft: Dict[str, Tensor]) -> Optional[Tensor]:
    """decoder.py:137-181 for enc_st_combine == 'none'.

    'dyn' with caption: vector = cat[query, cap, (spatial), (temporal)], scores =
    softmax(W vector); index 0 -> temporal, 1 -> spatial, 2 -> cap when both
    directions are on (decoder.py:148-159) -- note the order differs from the concat order.
    """
    D = "mutlimodal_decoder"
    v, c = cfg.nb_venc_blocks > 0, cfg.nb_cenc_blocks > 0
    if v and c and cfg.enc_vc_combine == "sum":
        assert cfg.s2t and cfg.t2s, "reference reads both keys (decoder.py:141)"
        return ft["temporal_ft"] + ft["spatial_ft"] + ft["cap_ft"]
    if v and c and cfg.enc_vc_combine == "dyn":
        parts = [ft["encoded_query"], ft["cap_ft"]]
        if cfg.t2s:
            parts.append(ft["spatial_ft"])
        if cfg.s2t:
            parts.append(ft["temporal_ft"])
        sc = torch.softmax(_lin(sd, f"{D}.vc_combine_W", torch.cat(parts, -1)), -1)
        if cfg.t2s and cfg.s2t:
            return sc[..., 0:1] * ft["temporal_ft"] + sc[..., 1:2] * ft["spatial_ft"] + sc[..., 2:3] * ft["cap_ft"]
        if not cfg.t2s:
            return sc[..., 0:1] * ft["temporal_ft"] + sc[..., 1:2] * ft["cap_ft"]
        return sc[..., 0:1] * ft["spatial_ft"] + sc[..., 1:2] * ft["cap_ft"]
    if v and not c and cfg.enc_vc_combine == "dyn":
        parts = [ft["encoded_query"]]
        if cfg.t2s:
            parts.append(ft["spatial_ft"])
        if cfg.s2t:
            parts.append(ft["temporal_ft"])
        sc = torch.softmax(_lin(sd, f"{D}.vc_combine_W", torch.cat(parts, -1)), -1)
        if cfg.t2s and cfg.s2t:
            return sc[..., 0:1] * ft["temporal_ft"] + sc[..., 1:2] * ft["spatial_ft"]
    return None


def mm_decoder(sd: SD, cfg: Cfg, b: OBatch, ft: Dict[str, Tensor], x: Tensor,
               trace: Optional[dict] = None) -> Dict[str, Tensor]:
    """MultimodalDecoder8.forward decoder.py:107-186."""
    D = "mutlimodal_decoder"
    q = ft["encoded_query"]
    in_ft = {"t2s": q, "s2t": q, "cap": q}
    for l in range(cfg.nb_blocks):
        if cfg.nb_venc_blocks > 0:
            tr = {} if trace is not None else None
            in_ft = vid_layer(sd, cfg, f"{D}.v_layers.{l}", in_ft, ft["spatiotemporal_ft"], b, tr)
            if trace is not None:
                trace[f"v{l}"] = tr
            if cfg.s2t:
                ft["temporal_ft"] = _ln(sd, f"{D}.temporal_out_norm", in_ft["s2t"])     # :127
            if cfg.t2s:
                ft["spatial_ft"] = _ln(sd, f"{D}.spatial_out_norm", in_ft["t2s"])       # :129
        if cfg.nb_cenc_blocks > 0:
            in_ft["cap"] = cap_layer(sd, cfg, f"{D}.c_layers.{l}", in_ft["cap"], ft["encoded_cap"], b)
            ft["cap_ft"] = _ln(sd, f"{D}.cap_out_norm", in_ft["cap"])                   # :132
        fused = fuse_modalities(sd, cfg, ft)
        if fused is not None:
            ft["encoded_ft"] = fused
        x = dec_layer(sd, cfg, f"{D}.layers.{l}", b, ft, x)                               # :182
    ft["decoded_text"] = _ln(sd, f"{D}.norm", x)                                         # :185
    return ft


# ----------------------------------------------------------------------------
# the model (model/mtn.py)
# ----------------------------------------------------------------------------
def mtn_encode(sd: SD, cfg: Cfg, b: OBatch) -> Dict[str, Tensor]:
    """MTN.encode mtn.py:36-51."""
    ft = encode_text(sd, cfg, b)
    if cfg.nb_venc_blocks > 0:
        ft["spatiotemporal_ft"] = vid_input_proj(sd, b.fts)
    return ft


def mtn_decode(sd: SD, cfg: Cfg, b: OBatch, ft: Dict[str, Tensor],
               trace: Optional[dict] = None) -> Dict[str, Tensor]:
    """MTN.decode mtn.py:53-61: target embedding is NOT layer-normed."""
    ft["encoded_tgt"] = embed(sd, b.trg, cfg.d_model)
    return mm_decoder(sd, cfg, b, ft, ft["encoded_tgt"], trace)


def mtn_forward(sd: SD, cfg: Cfg, b: OBatch, trace: Optional[dict] = None) -> Dict[str, Tensor]:
    """MTN.forward mtn.py:31-34."""
    cfg.check_scope()
    return mtn_decode(sd, cfg, b, mtn_encode(sd, cfg, b), trace)


# ----------------------------------------------------------------------------
# generator + loss (model/generator.py, label_smoothing.py, optimize.py)
# ----------------------------------------------------------------------------
def multi_pointer_generator(sd: SD, cfg: Cfg, ft: Dict[str, Tensor], b: OBatch) -> Tensor:
    """MultiPointerGenerator.forward generator.py:84-127 -> log-probs [B,Lt,V]."""
    x = ft["decoded_text"]
    lut = sd["generator.vocab_gen"]
    p_vocab = torch.softmax(x @ lut.t(), -1)
    names = cfg.ptr_ft.split(",")
    ptrs, vec = [], [x, ft["encoded_tgt"]]
    for idx, name in enumerate(names):
        text = {"query": b.query, "his": b.his, "cap": b.cap}[name]
        enc = ft["encoded_" + name]
        mask = {"query": b.query_mask, "his": b.his_mask, "cap": b.cap_mask}[name]
        if cfg.mask_unk:
            mask = mask & (text != UNK_ID).unsqueeze(-2)                      # generator.py:106-107
        _, p = mha(sd, f"generator.pointer_attn.{idx}", 1, x, enc, enc, mask)
        p = p.squeeze(1)                                                      # [B,Lt,Ltext]
        dist = torch.zeros_like(p_vocab).scatter_add(2, text.unsqueeze(1).expand_as(p), p)
        ptrs.append(dist)
        vec.append(p @ enc)                                                   # generator.py:117-118
    sw = torch.softmax(_lin(sd, "generator.pointer_gen_W", torch.cat(vec, -1)), -1)
    out = sw[..., -1:] * p_vocab
    for idx in range(len(names)):
        out = out + sw[..., idx:idx + 1] * ptrs[idx]
    return torch.log(out)


def ae_generator(sd: SD, ft: Dict[str, Tensor], key: str) -> Tensor:
    """Generator.forward generator.py:21-27 with the shared embedding matrix."""
    return torch.log_softmax(ft[key] @ sd["ae_generator.proj"].t(), -1)


def label_smoothing_kl(logp: Tensor, target: Tensor, size: int, smoothing: float = 0.1,
                       pad: int = PAD_ID) -> Tensor:
    """LabelSmoothing.forward label_smoothing.py:20-30: KLDiv(sum) against the smoothed one-hot."""
    conf = 1.0 - smoothing
    td = torch.full_like(logp, smoothing / (size - 2))
    td.scatter_(1, target.unsqueeze(1), conf)
    td[:, pad] = 0
    td = td * (target != pad).unsqueeze(1).to(td.dtype)
    # KLDivLoss(sum): sum td * (log td - logp), with 0*log0 := 0
    pos = td > 0
    return (td[pos] * (td[pos].log() - logp[pos])).sum()


def loss_compute(sd: SD, cfg: Cfg, ft: Dict[str, Tensor], b: OBatch, vocab: int) -> Dict[str, Tensor]:
    """SimpleLossCompute.__call__ optimize.py:46-83 (loss terms only; no optimiser step)."""
    out = multi_pointer_generator(sd, cfg, ft, b)
    norm = b.ntokens.float()
    qn = b.qntokens.float()
    losses = {"out": label_smoothing_kl(out.reshape(-1, vocab), b.trg_y.reshape(-1), vocab) / norm}
    total = losses["out"]
    if cfg.auto_encoder:
        keys = []
        if cfg.nb_cenc_blocks > 0:
            keys.append(("cap_ae", "cap_ft"))
        if cfg.nb_venc_blocks > 0:
            if cfg.s2t:
                keys.append(("temporal_ae", "temporal_ft"))
            if cfg.t2s:
                keys.append(("spatial_ae", "spatial_ft"))
        for name, key in keys:
            lp = ae_generator(sd, ft, key)
            losses[name] = label_smoothing_kl(lp.reshape(-1, vocab), b.query.reshape(-1), vocab) / qn
            total = total + losses[name]
    losses["total"] = total
    losses["logp"] = out
    return losses


# ----------------------------------------------------------------------------
# beam search (model/decode.py:53-104) -- host logic restated with the oracle model
# ----------------------------------------------------------------------------
def beam_search(sd: SD, cfg: Cfg, b: OBatch, max_len: int, beam: int = 5, penalty: float = 1.0,
                nbest: int = 5, min_len: int = 1, dec_eos: bool = False):
    """decode.py:53-104.  Batch size 1.  Ties are resolved by numpy argsort order exactly as the
    reference does (descending view of an ascending argsort)."""
    ft = mtn_encode(sd, cfg, b)
    hyps = [([], 0.0, torch.full((1, 1), SOS_ID, dtype=torch.long))]
    done, best = [], None
    for l in range(max_len):
        new, argmin = [], 0
        for out, lp, st in hyps:
            b.trg = st
            b.trg_mask = subsequent_mask(st.size(1))
            ft = mtn_decode(sd, cfg, b, ft)
            step = dict(ft)
            step["decoded_text"] = ft["decoded_text"][:, -1:]
            step["encoded_tgt"] = ft["encoded_tgt"][:, -1:]
            lp_vec = np.squeeze(multi_pointer_generator(sd, cfg, step, b).detach().numpy() + lp)
            if l >= min_len:
                new_lp = lp_vec[EOS_ID] + penalty * (len(out) + 1)
                done.append((out, new_lp))
                if best is None or best < new_lp:
                    best = new_lp
            for o in np.argsort(lp_vec)[::-1]:
                if o == UNK_ID or (not dec_eos and o == EOS_ID):
                    continue
                new_lp = lp_vec[o]
                if len(new) == beam:
                    if new[argmin][1] < new_lp:
                        new[argmin] = (out + [o], new_lp, torch.cat([st, torch.full((1, 1), int(o), dtype=torch.long)], 1))
                        argmin = min(enumerate(new), key=lambda e: e[1][1])[0]
                    else:
                        break
                else:
                    new.append((out + [o], new_lp, torch.cat([st, torch.full((1, 1), int(o), dtype=torch.long)], 1)))
                    if len(new) == beam:
                        argmin = min(enumerate(new), key=lambda e: e[1][1])[0]
        hyps = new
    if done:
        return sorted(done, key=lambda e: -e[1])[:nbest], best
    return [([], 0)], None


# ----------------------------------------------------------------------------
# deterministic parameters and inputs (shared by the golden generator and the tests)
# ----------------------------------------------------------------------------
def _rs(name: str) -> np.random.RandomState:
    import zlib
    return np.random.RandomState(zlib.crc32(name.encode()) & 0xFFFFFFFF)


def det_param(name: str, shape) -> Tensor:
    """A reproducible tensor for parameter ``name``: numpy's frozen RandomState stream seeded
    by crc32(name).  Matrices get a Xavier-like scale, biases 0.1, LayerNorm gains 1+0.1N."""
    shape = tuple(shape)
    z = _rs(name).standard_normal(shape).astype(np.float32)
    if name.endswith(".a_2"):
        z = 1.0 + 0.1 * z
    elif len(shape) == 1:
        z = 0.1 * z
    else:
        z = z * np.float32(math.sqrt(2.0 / (shape[0] + shape[1])))
    return torch.from_numpy(z)


def state_shapes(cfg: Cfg, vocab: int, feat_dim: int) -> Dict[str, tuple]:
    """Names and shapes of the reference model's parameters (verified against
    ``make_model(...).state_dict()`` by tests/golden/make_golden.py) minus the PE buffers."""
    d = cfg.d_model
    S: Dict[str, tuple] = {}

    def ln(p):
        S[p + ".a_2"] = (d,); S[p + ".b_2"] = (d,)

    def lin(p, o, i):
        S[p + ".weight"] = (o, i); S[p + ".bias"] = (o,)

    def attn(p):
        for j in range(4):
            lin(f"{p}.linears.{j}", d, d)

    def ff(p):
        lin(p + ".w_1", 4 * d, d); lin(p + ".w_2", d, 4 * d)

    for j in range(3):
        ln(f"text_encoder.norm.{j}")
    lin("vid_encoder.W", d, feat_dim)
    ln("vid_encoder.in_norm")
    D = "mutlimodal_decoder"
    both = cfg.t2s and cfg.s2t
    nva, nvf = (6, 2) if both else (3, 1)
    nda = 3 + (1 if (cfg.nb_cenc_blocks > 0 and cfg.nb_venc_blocks > 0 and cfg.enc_vc_combine != "none") else
               ((1 if cfg.nb_cenc_blocks > 0 else 0) + (2 if cfg.nb_venc_blocks > 0 else 0)))
    for l in range(cfg.nb_blocks):
        p = f"{D}.layers.{l}"
        for j in range(nda):
            attn(f"{p}.attn.{j}")
        ff(f"{p}.ff")
        for j in range(nda + 1):
            ln(f"{p}.sublayer.{j}.norm")
    ln(f"{D}.norm")
    for l in range(cfg.nb_venc_blocks):
        p = f"{D}.v_layers.{l}"
        for j in range(nva):
            attn(f"{p}.attn.{j}")
        for j in range(nvf):
            ff(f"{p}.ff.{j}")
        for j in range(nva + nvf):
            ln(f"{p}.sublayer.{j}.norm")
    if cfg.nb_venc_blocks > 0:
        ln(f"{D}.spatial_out_norm"); ln(f"{D}.temporal_out_norm")
    for l in range(cfg.nb_cenc_blocks):
        p = f"{D}.c_layers.{l}"
        attn(f"{p}.attn.0"); attn(f"{p}.attn.1"); ff(f"{p}.ff")
        for j in range(3):
            ln(f"{p}.sublayer.{j}.norm")
    if cfg.nb_cenc_blocks > 0:
        ln(f"{D}.cap_out_norm")
    if cfg.nb_venc_blocks > 0 and cfg.enc_vc_combine == "dyn":
        factor = 1 + (1 if cfg.include_caption != "none" else 0) + (1 if cfg.t2s else 0) + (1 if cfg.s2t else 0)
        lin(f"{D}.vc_combine_W", factor - 1, d * factor)
    S["query_embed.0.lut.weight"] = (vocab, d)
    n_ptr = len(cfg.ptr_ft.split(","))
    for j in range(n_ptr):
        attn(f"generator.pointer_attn.{j}")
    lin("generator.pointer_gen_W", n_ptr + 1, d * (n_ptr + 2))
    return S


def det_state(cfg: Cfg, vocab: int, feat_dim: int) -> SD:
    """Deterministic full state dict (with the shared-embedding aliases filled in)."""
    sd = {k: det_param(k, s) for k, s in state_shapes(cfg, vocab, feat_dim).items()}
    lut = sd["query_embed.0.lut.weight"]
    sd["tgt_embed.0.lut.weight"] = lut
    sd["generator.vocab_gen"] = lut
    sd["ae_generator.proj"] = lut
    return sd


def det_batch(B: int, T: int, S: int, C: int, Lq: int, Lh: int, Lc: int, Lt: int, vocab: int,
              seed: int = 1234, ragged: bool = True, fully_masked_clip: bool = False) -> OBatch:
    """Synthetic batch per SURVEY.md 8(d): N(0,1) features with trailing all-zero temporal rows
    (which is what drives temporal_mask), ids in [4,V), trailing pad on some rows."""
    rs = np.random.RandomState(seed)
    fts = rs.standard_normal((B, T, S, C)).astype(np.float32)
    if ragged:
        for i in range(B):
            tb = int(rs.randint(max(1, T // 2), T + 1))
            fts[i, tb:] = 0.0
    if fully_masked_clip:
        fts[B - 1] = 0.0

    def ids(L, pad_some):
        x = rs.randint(4, vocab, size=(B, L)).astype(np.int64)
        if pad_some:
            for i in range(B):
                if rs.rand() < 0.5 and L > 2:
                    x[i, int(rs.randint(L // 2, L)):] = PAD_ID
        return torch.from_numpy(x)

    query, his, cap = ids(Lq, ragged), ids(Lh, ragged), ids(Lc, ragged)
    query[0, 1] = UNK_ID              # exercise mask_unk in the pointer generator
    full = ids(Lt + 1, False)
    full[:, 0] = SOS_ID
    trg, trg_y = full[:, :-1].clone(), full[:, 1:].clone()
    if ragged:
        for i in range(B):
            if rs.rand() < 0.5 and Lt > 2:
                cut = int(rs.randint(Lt // 2, Lt))
                trg_y[i, cut:] = PAD_ID
                trg[i, cut + 1:] = PAD_ID
                trg_y[i, cut - 1] = EOS_ID
    return OBatch(query=query, his=his, cap=cap, trg=trg, trg_y=trg_y, fts=torch.from_numpy(fts))
