"""The TRAINING form of the fused stage-1 launch (bist_st_stage1_fused_train_fwd: value projection, scores, masked softmax + dropout,
P.V, output projection + dropout + residual in one launch, V / probabilities / context rows as side outputs) and the backward core fed
with the saved probabilities (bist_st_stage1_pv_bwd_p), against the unfused kernels they replace -- same dropout seeds, so the two
paths draw the SAME masks (probability mask index ((((b G + g) h + hh) Lq + i) K + key, output mask index row * d + column) -- and,
through the model, against the unfused training step."""
import argparse
import math

import pytest
import torch

from oracle import bist_oracle as O

pytestmark = pytest.mark.gpu


def _args(cfg):
    return argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import bist_amd.model as M
    from bist_amd.data.batch import Batch
    return M, Batch


def _operands(B, T, S, Lq, seed):
    g = torch.Generator().manual_seed(seed)
    d, h = 512, 8
    bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).cuda()
    vft = bf(B, T, S, d)
    qf = bf(B, Lq * h, d, sc=0.05)
    x = bf(B, Lq, d)
    wv, wo = bf(d, d, sc=d ** -0.5), bf(d, d, sc=d ** -0.5)
    bv, bo = bf(d, sc=0.1), bf(d, sc=0.1)
    return vft, qf, x, wv, bv, wo, bo, d, h


@pytest.mark.parametrize("direction,T,S,Lq,masked", [(0, 32, 49, 20, True), (1, 32, 49, 20, False), (0, 128, 9, 20, True), (1, 8, 49, 12, False),
                                                      (0, 20, 7, 20, True), (1, 6, 33, 20, False)])
@pytest.mark.parametrize("drop", [False, True], ids=["eval", "dropout"])
def test_fused_training_forward_matches_the_unfused_kernels(direction, T, S, Lq, masked, drop):
    from bist_amd import functional as Fn, ops
    B = 3
    vft, qf, x, wv, bv, wo, bo, d, h = _operands(B, T, S, Lq, 100 * direction + T + S)
    G, K = (S, T) if direction == 0 else (T, S)
    assert ops.st_stage1_fused_train_ok(T, S, Lq, d, h, direction, torch.bfloat16)
    kmask = None
    if masked:
        kmask = torch.ones(B, K, dtype=torch.uint8)
        kmask[0, K // 2:] = 0
        kmask[1, :] = 0                      # a fully masked clip: uniform probabilities (modules.py:60)
        kmask = kmask.cuda()
    adrop = (0.1, Fn.next_seed()) if drop else None
    sdrop = (0.1, Fn.next_seed()) if drop else None
    wvf, wof = ops.pack_frag_rows(wv), ops.pack_frag_rows(wo)
    y, v, p, o = ops.st_stage1_fused_train(qf, vft, kmask, wvf, bv, wof, bo, x, h=h, direction=direction, attn_drop=adrop, sub_drop=sdrop)
    # the unfused kernels on the same operands and seeds
    v_ref = ops.linear(vft.view(B * T * S, d), wv, bv).view(B, T, S, d)
    sc = torch.empty(B, Lq * h, T * S, device="cuda", dtype=torch.float32)
    vf = vft.view(B, T * S, d)
    ops.gemm(qf, vf, sc, M=Lq * h, N=T * S, K=d, a_rs=d, b_rs=d, ldc=T * S, batch=(B, 1), a_bs=(Lq * h * d, 0), b_bs=(T * S * d, 0), c_bs=(Lq * h * T * S, 0))
    o_ref = ops.st_stage1_pv(sc, v_ref, kmask, B=B, T=T, S=S, Lq=Lq, h=h, dk=d // h, direction=direction, drop=adrop)
    sp, ss = sdrop if sdrop else (0.0, 0)
    y_ref = ops.linear(o_ref.view(B * G * Lq, d), wo, bo, residual=x.view(B * Lq, d), res_map=(G * Lq, Lq), drop_p=sp, drop_seed=ss).view(B, G, Lq, d)
    torch.cuda.synchronize()
    assert (v.float() - v_ref.float()).abs().max().item() <= 4e-2                       # one bf16 rounding step of O(1) values
    # probabilities: softmax of the scores with the reference's masking rule
    s5 = sc.view(B, Lq, h, T, S)
    s5 = s5.permute(0, 4, 2, 1, 3) if direction == 0 else s5.permute(0, 3, 2, 1, 4)   # [B, G, h, Lq, K]
    if kmask is not None:
        s5 = s5.masked_fill(kmask.view(B, 1, 1, 1, K) == 0, -1e9)
    p_ref = torch.softmax(s5.float(), dim=-1)
    assert (p[..., :K] - p_ref).abs().max().item() <= 2e-3
    assert p.shape[-1] % 4 == 0 and (p[..., K:] == 0).all()
    assert (o.float() - o_ref.float()).abs().max().item() <= 6e-2, "context rows (the attention dropout masks of the two paths must coincide)"
    assert (y.float() - y_ref.float()).abs().max().item() <= 1.2e-1, "outputs (the sublayer dropout masks of the two paths must coincide)"
    if drop:      # the masks really dropped something, and identically: zeros of drop(W_o ctx + b_o) sit at the same places
        za, zb = (y.float() - x.float().view(B, 1, Lq, d)) == 0, (y_ref.float() - x.float().view(B, 1, Lq, d)) == 0
        assert 0.05 < za.float().mean().item() < 0.15 and (za == zb).float().mean().item() > 0.99      # (bf16: x + tiny rounds to x in either path)


@pytest.mark.parametrize("own_v", [True, False], ids=["own_values", "values_given"])
@pytest.mark.parametrize("direction,T,S", [(0, 32, 49), (1, 32, 49), (0, 128, 9)])
def test_fused_training_node_gradients_match_the_unfused_autograd_path(direction, T, S, own_v):
    """St1FusedTrainFn (one launch forward; backward on the saved V / probabilities / context) against the four-node unfused path
    (value projection, score product, softmax + P.V core, output projection) under autograd, both dropouts on, same seeds."""
    from bist_amd import functional as Fn, ops
    from bist_amd.model.modules import MultiHeadedAttention
    B, Lq = 2, 20
    vft, qf, x, wv, bv, wo, bo, d, h = _operands(B, T, S, Lq, 7 + direction)
    G, K = (S, T) if direction == 0 else (T, S)
    attn = MultiHeadedAttention(h, d, dropout=0.1).cuda().to(torch.bfloat16)
    with torch.no_grad():
        attn.linears[2].weight.copy_(wv); attn.linears[2].bias.copy_(bv); attn.linears[3].weight.copy_(wo); attn.linears[3].bias.copy_(bo)
    tmask = None
    if direction == 0:
        tmask = torch.ones(B, 1, K, dtype=torch.bool)
        tmask[0, 0, K - 5:] = False
        tmask = tmask.cuda()
    adrop, sdrop = (0.1, Fn.next_seed()), (0.1, Fn.next_seed())
    cot = (torch.randn(B, G, Lq, d, generator=torch.Generator().manual_seed(3)) * 0.1).to(torch.bfloat16).cuda()
    res = {}
    for fused in (True, False):
        leaves = [t.detach().clone().requires_grad_(True) for t in (qf, x, vft)]
        q_, x_, v_ = leaves
        attn.zero_grad(set_to_none=True)
        if fused:
            frag = (ops.pack_frag_rows(attn.linears[2].weight.detach()), ops.pack_frag_rows(attn.linears[3].weight.detach()))
            if own_v:       # the launch projects (and saves) the values itself
                y, _x2 = Fn.st_stage1_fused_train(q_, x_, v_, v_, tmask, attn, frag, h=h, direction=direction, attn_drop=adrop, sub_drop=sdrop)
            else:           # the value projection stays a product of its own (the default: it runs off the direction's chain)
                val = Fn.linear(v_.view(B * T * S, d), attn.linears[2].weight, attn.linears[2].bias).view(B, T, S, d)
                y, _x2 = Fn.st_stage1_fused_train(q_, x_, v_, None, tmask, attn, frag, h=h, direction=direction, attn_drop=adrop, sub_drop=sdrop, v=val)
        else:
            val = Fn.linear(v_.view(B * T * S, d), attn.linears[2].weight, attn.linears[2].bias).view(B, T, S, d)
            sc = Fn.st_scores(q_, v_.view(B, T * S, d))
            o = Fn.st_stage1_pv(sc, val, tmask, B=B, T=T, S=S, Lq=Lq, h=h, dk=d // h, direction=direction, drop=adrop)
            y = Fn.linear(o, attn.linears[3].weight, attn.linears[3].bias, residual=x_, res_map=(G * Lq, Lq), drop_p=sdrop[0], drop_seed=sdrop[1]).view(B, G, Lq, d)
        (y.float() * cot.float()).sum().backward()
        torch.cuda.synchronize()
        res[fused] = [y.detach().float()] + [t.grad.float() for t in leaves] + [p.grad.float() for p in (attn.linears[2].weight, attn.linears[2].bias,
                                                                                                        attn.linears[3].weight, attn.linears[3].bias)]
    names = ["y", "dqf", "dx", "dvft", "dWv", "dbv", "dWo", "dbo"]
    for n, a, b in zip(names, res[True], res[False]):
        scale = max(b.abs().max().item(), 1e-6)
        rel = (a - b).abs().max().item() / scale
        cos = torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
        assert cos >= 0.998 and rel <= 6e-2, (n, cos, rel)


def test_training_step_with_the_fused_stage1_matches_the_unfused_step(hip):
    """Two trainers on identical models and batches, train() mode with dropout 0.1 at all four sites and the same seed stream: stage 1 as
    the fused training launch (asserted through bist_launch_count) against the four-launch form.  Same masks, same losses to bf16
    rounding, and the same weights after three optimiser steps to within what bf16 storage of activations allows."""
    from bist_amd import _lib, functional as Fn
    from bist_amd.train import Trainer
    M, Batch = hip
    cfg = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2, dropout=0.1)
    V, C = 300, 256
    ob = O.det_batch(4, 32, 49, C, 20, 30, 15, 12, V, seed=23)
    out = {}
    old = Fn.FUSED_TRAIN
    try:
        for fused in (True, False):
            Fn.FUSED_TRAIN = fused
            Fn.manual_seed(99)
            model = M.make_model(V, V, _args(cfg), ft_sizes=[C])
            model.load_state_dict(O.det_state(cfg, V, C), strict=False)
            model = model.to("cuda").to(torch.bfloat16).train()
            tr = Trainer(model, _args(cfg), V, compute_dtype=torch.bfloat16, warmup=10, use_graph=False)
            b = Batch(ob.query.cuda(), ob.his.cuda(), ob.fts.cuda().to(torch.bfloat16), ob.cap.cuda(), ob.trg.cuda(), ob.trg_y.cuda())
            _lib.lib.bist_launch_count_reset()
            losses = [{k: float(v) for k, v in tr.step(b).items()} for _ in range(3)]
            torch.cuda.synchronize()
            n_f, n_p, n_u = (_lib.lib.bist_launch_count(k) for k in (_lib.K_ST1_FUSED_TRAIN, _lib.K_ST1_PBWD, _lib.K_ST1_MFMA_FWD))
            assert (n_f, n_p, n_u) == ((12, 12, 0) if fused else (0, 0, 12)), (fused, n_f, n_p, n_u)
            out[fused] = (losses, tr.master.clone())
    finally:
        Fn.FUSED_TRAIN = old
    for la, lb in zip(out[True][0], out[False][0]):
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-2 * max(1.0, abs(lb[k])), (k, la, lb)
    wa, wb = out[True][1].double(), out[False][1].double()
    assert torch.nn.functional.cosine_similarity(wa, wb, dim=0).item() > 0.9999
