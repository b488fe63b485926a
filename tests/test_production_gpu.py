"""The bf16 PRODUCTION kernels at the bench geometry against the CPU oracle, end to end.

BASELINE configs[1] / configs[3] geometry: d_model=512, h=8, C=2048, S=49, Lq=20, B=16 clips (so the frame-grid products are the
256x256-tile kernel's, bist_gemm_is_fast == 4), T=32 and T=128, two layers (the oracle, fp32 on the host cores, takes seconds).
The oracle gets the SAME bf16-rounded weights and features as fp32 values, so what is measured is the path's bf16 arithmetic
(storage rounding of activations, fp32 accumulation), not the rounding of the inputs.  Each test also states WHICH kernels ran
(bist_launch_count): the fused stage-1 kernel in inference, the matrix-core stage-1 / stage-2 / small-attention-backward kernels
under autograd -- never the fp32 VALU fallbacks.

Bounds (bf16: 8 significant bits): layer-normed activations and log-probs within 6e-2 absolute of the oracle (measured ~1-2e-2),
>= 90 % identical greedy argmax, losses within 2 %; gradients: cosine >= 0.995 and max error <= 8 % of the tensor's largest entry.
"""
import argparse

import pytest
import torch

from oracle import bist_oracle as O

pytestmark = pytest.mark.gpu

V, C, S, LQ, LH, LC, LT, B = 3000, 2048, 49, 20, 60, 25, 20, 16
GRAD_KEYS = ["vid_encoder.W.weight", "vid_encoder.in_norm.a_2", "query_embed.0.lut.weight", "generator.pointer_gen_W.weight",
             "mutlimodal_decoder.vc_combine_W.weight", "mutlimodal_decoder.layers.1.attn.0.linears.0.weight",
             "mutlimodal_decoder.c_layers.0.attn.1.linears.2.weight"] + \
            [f"mutlimodal_decoder.v_layers.{l}.attn.{ai}.linears.{j}.weight" for l in (0, 1) for ai in range(6) for j in (0, 1, 2, 3)] + \
            [f"mutlimodal_decoder.v_layers.0.ff.{f}.w_{k}.weight" for f in (0, 1) for k in (1, 2)] + \
            [f"mutlimodal_decoder.v_layers.0.sublayer.{si}.norm.a_2" for si in range(8)]


def _args(cfg):
    return argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})


@pytest.fixture(scope="module", params=[32, 128], ids=["T32", "T128"])
def prod(request):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import bist_amd.model as M
    from bist_amd.data.batch import Batch
    T = request.param
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)
    aliases = ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj")      # one shared tensor (mtn.py:82,90,93)
    sd = {k: v.bfloat16().float() for k, v in O.det_state(cfg, V, C).items()}             # bf16-representable weights
    for a in aliases:
        sd[a] = sd["query_embed.0.lut.weight"]
    ob = O.det_batch(B, T, S, C, LQ, LH, LC, LT, V, seed=7)
    ob.fts = ob.fts.bfloat16().float()
    # oracle: forward, losses, gradients (fp32, host cores)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in aliases}
    for a in aliases:
        leaf[a] = leaf["query_embed.0.lut.weight"]
    ref = O.mtn_forward(leaf, cfg, ob)
    ref_logp = O.multi_pointer_generator(leaf, cfg, ref, ob)
    losses = O.loss_compute(leaf, cfg, ref, ob, V)
    losses["total"].backward()
    ref_grads = {k: leaf[k].grad.detach().clone() for k in GRAD_KEYS}
    ref = {k: v.detach() for k, v in ref.items()}
    model = M.make_model(V, V, _args(cfg), ft_sizes=[C])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(".pe") for k in missing)
    model = model.to("cuda").to(torch.bfloat16).eval()
    b = Batch(ob.query.cuda(), ob.his.cuda(), ob.fts.cuda().to(torch.bfloat16), ob.cap.cuda(), ob.trg.cuda(), ob.trg_y.cuda())
    return dict(T=T, cfg=cfg, model=model, b=b, ref=ref, ref_logp=ref_logp.detach(), losses={k: float(v.detach()) for k, v in losses.items() if k != "logp"},
                ref_grads=ref_grads)


def _err(a, b):
    return (a.detach().float().cpu().double() - b.double()).abs().max().item()


def _counts():
    from bist_amd import _lib
    return {n: _lib.lib.bist_launch_count(getattr(_lib, n)) for n in dir(_lib) if n.startswith("K_")}


def _check_forward(ft, logp, p, what):
    worst = {k: _err(ft[k], p["ref"][k]) for k in p["ref"] if k in ft and torch.is_tensor(ft[k])}
    worst["logp"] = _err(logp, p["ref_logp"])
    bad = {k: v for k, v in worst.items() if not v <= 6e-2}
    assert not bad, f"{what}: beyond the bf16 bound of the oracle: {bad} (all: {worst})"
    agree = (logp.argmax(-1).cpu() == p["ref_logp"].argmax(-1)).float().mean().item()
    assert agree >= 0.9, (what, agree)
    for k in ("spatiotemporal_ft", "temporal_ft", "spatial_ft", "cap_ft", "encoded_ft", "decoded_text"):
        assert k in worst, k


def test_p0_takes_the_256_tile_kernel(prod):
    from bist_amd import _lib, ops
    p = prod
    M_ = B * p["T"] * S
    x = torch.empty(M_, C, device="cuda", dtype=torch.bfloat16)
    w = p["model"].vid_encoder.W.weight
    g = ops.gemm_desc(x, w, torch.empty(M_, 512, device="cuda", dtype=torch.bfloat16), M=M_, N=512, K=C, a_rs=C, b_rs=C, ldc=512)
    assert _lib.lib.bist_gemm_is_fast(g) == 4


def test_inference_path_matches_oracle(prod):
    """torch.no_grad(): stage 1 of both directions is bist_st_stage1_fused_fwd, stage 2 the matrix-core kernel."""
    from bist_amd import _lib
    p = prod
    _lib.lib.bist_launch_count_reset()
    with torch.no_grad():
        ft = p["model"].forward(p["b"])
        logp = p["model"].generator(ft, p["b"], _args(p["cfg"]))
    torch.cuda.synchronize()
    c = _counts()
    assert c["K_ST1_FUSED"] == 4 and c["K_ST2_MFMA_FWD"] == 4, c          # 2 layers x 2 directions
    assert c["K_ST1_VALU"] == 0 and c["K_ST2_VALU"] == 0 and c["K_ST1_MFMA_FWD"] == 0, c
    _check_forward(ft, logp, p, f"inference T={p['T']}")


def test_training_path_matches_oracle_forward_losses_and_gradients(prod):
    """Autograd on (eval mode: dropout off, as in the oracle): the forward goes through the kernels the bench's training step runs
    (stage 1 as the fused training launch bist_st_stage1_fused_train_fwd, st2_mfma), the backward through bist_st_stage1_pv_bwd_p, the
    stage-2 backward and mha_bwd_mfma."""
    from bist_amd import _lib
    from bist_amd.model.label_smoothing import LabelSmoothing
    from bist_amd.model.optimize import SimpleLossCompute
    p = prod
    model, b = p["model"], p["b"]
    model.zero_grad(set_to_none=True)
    _lib.lib.bist_launch_count_reset()
    lc = SimpleLossCompute(model.generator, model.ae_generator, LabelSmoothing(V, O.PAD_ID, 0.1), None, args=_args(p["cfg"]))
    ft = model.forward(b)
    terms, logp = lc.terms(ft, b)
    total = sum(terms.values())
    total.backward()
    torch.cuda.synchronize()
    c = _counts()
    # stage 1: forward = the fused TRAINING launch (one per direction and layer), backward = the matrix-core core fed with its saved
    # probabilities; stage 2 and the small attentions on their matrix-core kernels; never the fp32 VALU fallbacks
    assert c["K_ST1_FUSED_TRAIN"] == 4 and c["K_ST1_PBWD"] == 4 and c["K_ST2_MFMA_FWD"] == 4 and c["K_ST2_MFMA_BWD"] == 4, c
    assert c["K_ST1_MFMA_FWD"] == 0 and c["K_ST1_MFMA_BWD"] == 0 and c["K_ST1_FUSED"] == 0, c
    assert c["K_MHA_BWD_MFMA"] > 0 and c["K_ST1_VALU"] == 0 and c["K_ST2_VALU"] == 0 and c["K_MHA_BWD_VALU"] == 0, c
    _check_forward({k: v for k, v in ft.items()}, logp, p, f"training forward T={p['T']}")
    for name, val in terms.items():
        ref = p["losses"][name]
        assert abs(val.item() - ref) <= 2e-2 * max(1.0, abs(ref)), (name, val.item(), ref)
    sd = dict(model.named_parameters())
    worst = {}
    for k in GRAD_KEYS:
        ref = p["ref_grads"][k].double()
        got = sd[k].grad.detach().float().cpu().double()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        rel = ((got - ref).abs().max() / max(1e-8, ref.abs().max())).item()
        if not (cos >= 0.995 and rel <= 8e-2):
            worst[k] = (cos, rel)
    assert not worst, f"T={p['T']}: gradient (cosine, max relative error) out of the bf16 bound: {worst}"
