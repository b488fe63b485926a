"""``MTN`` container and ``make_model`` factory with the reference's signatures and parameter
naming (reference: model/mtn.py) so that train.py / generate.py call it unchanged.

``state_dict`` names are identical to the reference's (including its spelling
``mutlimodal_decoder``), the embedding matrix is one tensor shared by ``query_embed``,
``tgt_embed``, ``generator.vocab_gen`` and ``ae_generator.proj`` (mtn.py:82,90,101).
"""
from __future__ import annotations

import copy
import os
from typing import Dict

import torch
import torch.nn as nn

from .decoder import MultimodalDecoder8, MultimodalDecoderLayer12
from .encoder import AudioEncoderLayer, CapEncoderLayer, Encoder, VidEncoder8, VidEncoderLayer4
from .generator import Generator, MultiPointerGenerator, PointerGenerator
from .modules import (Embeddings, MultiHeadedAttention, PositionalEncoding, PositionwiseFeedForward,
                      embed_with_position)

from .. import functional as Fn
from .. import stamps as STM

TGT_EMBED_ON_DECODER_STREAM = os.environ.get("BIST_TGT_EMBED_SIDE", "1") != "0"      # tuning aid, see MTN.multimodal_decode_text

Tensor = torch.Tensor


class MTN(nn.Module):
    """forward(b) -> ft dict with the reference's keys (mtn.py:17-61)."""

    def __init__(self, args, text_encoder, vid_encoder, text_decoder, mutlimodal_decoder, query_embed, his_embed,
                 cap_embed, tgt_embed, generator, ae_generator, ptr_gen=False):
        super().__init__()
        self.text_encoder = text_encoder
        self.vid_encoder = vid_encoder
        self.text_decoder = text_decoder
        self.mutlimodal_decoder = mutlimodal_decoder
        self.query_embed = query_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.ae_generator = ae_generator
        self.ptr_gen = ptr_gen
        self.args = args

    _bist_param_gates = True      # this model calls Fn.param_gate before the first read of every parameter piece (bist_amd/train.py)

    def _flush_trainer(self) -> None:
        """A trainer with the deferred optimiser leaves the last step's update pending: apply it before the weights are read."""
        ref = self.__dict__.get("_bist_trainer")
        tr = ref() if ref is not None else None
        if tr is not None:
            tr.flush()

    def __getstate__(self):
        """torch.save(model) / copy.deepcopy(model): flush a pending update, and leave this build's runtime state behind -- the trainer
        reference, the captured decode graphs and their staleness key, the cached module list (instance attributes named _bist_*: all of
        them per-process or derived, rebuilt on demand)."""
        self._flush_trainer()
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_bist_")}

    def train(self, mode: bool = True):
        if not mode:
            self._flush_trainer()
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        self._flush_trainer()
        return super().state_dict(*args, **kwargs)

    def forward(self, b) -> Dict[str, Tensor]:
        return self.decode(b, self.encode(b))

    def encode(self, b) -> Dict[str, Tensor]:
        if torch.is_grad_enabled():
            Fn.release_taken()        # a training loop without bist_amd.train.Trainer: the previous step's cross-stream tensors may go now
        return self.encode_vid(b, self.encode_text(b, {}))

    def encode_text(self, b, ft):
        Fn.param_gate(0)
        e = self.query_embed
        q, c, h = self.text_encoder(embed_with_position(e, b.query),
                                    embed_with_position(e, b.cap) if b.cap is not None else None,
                                    embed_with_position(e, b.his))
        ft["encoded_query"], ft["encoded_cap"], ft["encoded_his"] = STM.through(q, "text enc"), c, h
        return ft

    def encode_vid(self, b, ft):
        ft = self.vid_encoder(b, ft)
        if STM.ENABLED and "spatiotemporal_ft" in ft:
            ft["spatiotemporal_ft"] = STM.through(ft["spatiotemporal_ft"], "P0")
        return ft

    def decode(self, b, ft, pos0: int = 0):
        """pos0 (beam search, one decode step at a time): ``b.trg`` holds only the tokens at positions pos0 .. of the prefixes."""
        return self.multimodal_decode_text(b, ft, pos0)

    def multimodal_decode_text(self, b, ft, pos0: int = 0):
        if TGT_EMBED_ON_DECODER_STREAM and Fn.KEEP_TAKEN and torch.is_grad_enabled() and Fn.CONCURRENT and Fn.PIPELINE_DECODER and b.trg.is_cuda:
            # Training: the target embedding on the stream of the decoder layers.  Its gradient's last addend comes from decoder layer 0 at the
            # very end of that stream's backward chain; as a main-stream node, autograd's accumulation made the MAIN stream wait there -- with
            # everything the engine issued on it afterwards (the text encoders' and the input projection's backward) -- for ~0.4 ms.
            main, side = torch.cuda.current_stream(), Fn.side_stream(1)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ft["encoded_tgt"] = embed_with_position(self.tgt_embed, b.trg, pos0)
            Fn._keep_taken(ft["encoded_tgt"])             # (allocated on that stream; the generator reads it on the main one)
        else:
            ft["encoded_tgt"] = embed_with_position(self.tgt_embed, b.trg, pos0)     # not layer-normed (mtn.py:58-59)
        return self.mutlimodal_decoder(b, ft, ft["encoded_tgt"])


def make_model(src_vocab, tgt_vocab, args, ft_sizes=None, embeddings=None):
    """Same construction order and sharing as the reference factory (mtn.py:63-167): d_ff is always
    4*d_model (args.d_ff is ignored there too), every attention/ff is an independent deep copy,
    xavier-uniform on every parameter with more than one dimension."""
    N, venc_N, cenc_N, aenc_N = args.nb_blocks, args.nb_venc_blocks, args.nb_cenc_blocks, args.nb_aenc_blocks
    d_model, h, dropout = args.d_model, args.att_h, args.dropout
    d_ff = d_model * 4
    c = copy.deepcopy
    attn = MultiHeadedAttention(h, d_model)
    ff = PositionwiseFeedForward(d_model, d_ff, dropout)
    query_embed = nn.Sequential(Embeddings(d_model, src_vocab), PositionalEncoding(d_model, dropout))
    tgt_embed = query_embed
    lut = tgt_embed[0].lut.weight
    if not args.ptr_gen:
        raise NotImplementedError("ptr_gen=0 raises NameError in the reference (mtn.py:95); only ptr_gen=1 is in scope")
    names = args.ptr_ft.split(",")
    if len(names) > 1:
        ptr = nn.ModuleList(MultiHeadedAttention(1, d_model, dropout=0) for _ in names)
        generator = MultiPointerGenerator(d_model, lut, ptr, len(names))
    else:
        generator = PointerGenerator(d_model, lut, MultiHeadedAttention(1, d_model, dropout=0))
    ae_generator = Generator(d_model, tgt_vocab, lut) if args.auto_encoder else None
    text_encoder = Encoder(d_model, nb_layers=3)
    if not ft_sizes:
        raise ValueError("ft_sizes must list the video feature width (mtn.py:110-120)")
    vid_W = nn.Linear(ft_sizes[0], d_model)
    a_W = nn.Linear(ft_sizes[1], d_model) if len(ft_sizes) > 1 else None
    vid_encoder = VidEncoder8(c(vid_W), c(a_W), None, venc_N, aenc_N, d_model, args)
    both = bool(args.t2s) and bool(args.s2t)
    v_layer = VidEncoderLayer4(d_model, c(attn), 6 if both else 3, c(ff), 2 if both else 1, dropout, args)
    c_layer = CapEncoderLayer(d_model, c(attn), 2, c(ff), dropout)
    a_layer = AudioEncoderLayer(d_model, c(attn), 2, c(ff), dropout)
    nb_attn = 3
    if cenc_N > 0 and venc_N > 0 and args.enc_vc_combine != "none":
        nb_attn += 1
    else:
        nb_attn += (cenc_N > 0) + (aenc_N > 0)
        if venc_N > 0:
            nb_attn += 1 if (args.enc_st_combine in ("dyn", "sum", "early_sum", "early_dyn") and both) else 2
    mm_layer = MultimodalDecoderLayer12(d_model, c(attn), nb_attn, c(ff), dropout, args)
    decoder = MultimodalDecoder8(v_layer, c_layer, a_layer, mm_layer, venc_N, cenc_N, aenc_N, N, args)
    model = MTN(args=args, text_encoder=text_encoder, vid_encoder=vid_encoder, text_decoder=None,
                mutlimodal_decoder=decoder, query_embed=query_embed, his_embed=None, cap_embed=None,
                tgt_embed=tgt_embed, generator=generator, ae_generator=ae_generator, ptr_gen=args.ptr_gen)
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
