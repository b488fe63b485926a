"""The lock-step form of the reasoning layer (bist_amd/zbatch.py: both directions' query-side operations as ONE sequence of
launches over stacked [2, B, Lq, d] tensors) against (1) the reference's per-stage golden outputs, (2) the two-chain form of the
same layer (round 2; itself checked against the oracle) -- forward and every parameter gradient -- and (3) torch autograd with the
kernels' own dropout masks recovered (the z-indexed mask of the product epilogue and the masked-gradient hand-off through the
LayerNorm backward)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import bist_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import bist_amd.model as M
    from bist_amd.data.batch import Batch
    return M, Batch


def _args(cfg):
    return argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})


def _model(M, cfg, V, C, dtype):
    model = M.make_model(V, V, _args(cfg), ft_sizes=[C])
    sd = O.det_state(cfg, V, C)
    model.load_state_dict(sd, strict=False)
    return model.to("cuda").to(dtype).eval(), sd


def _batch(Batch, ob, dtype):
    return Batch(ob.query.cuda(), ob.his.cuda(), ob.fts.cuda().to(dtype), ob.cap.cuda(), ob.trg.cuda(), ob.trg_y.cuda())


@pytest.mark.parametrize("tag", ["both", "mid", "masked"])
@pytest.mark.parametrize("zbatch", [True, False], ids=["lockstep", "two_chains"])
def test_per_stage_outputs_match_the_reference_golden(hip, golden_dir, tag, zbatch):
    """VidEncoderLayer4's sublayer outputs A0, A1, A2, F0, A3, A4, A5, F1 of layer 0 (encoder.py:121,130,135,148,161,166 through the
    reference's own SublayerConnection hooks, tests/golden/g3_model.npz `*_v0_sublayer{i}`) against the HIP path's per-stage tensors
    (debug hook `_bist_trace`), fp32, 1e-3 -- for the lock-step form and the two-chain form."""
    from bist_amd import zbatch as Z
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g3_model.npz"))
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm.get("seed", 1234),
                     fully_masked_clip=meta.get("fully_masked", False))
    model, _ = _model(M, cfg, dm["V"], dm["C"], torch.float32)
    layer = model.mutlimodal_decoder.v_layers[0]
    trace = layer.__dict__["_bist_trace"] = {}
    old = Z.ENABLED
    Z.ENABLED = zbatch
    try:
        with torch.no_grad():
            model.forward(_batch(Batch, ob, torch.float32))
    finally:
        Z.ENABLED = old
        layer.__dict__.pop("_bist_trace", None)
    names = ["t2s_self", "t2s_stage1", "t2s_stage2", "t2s_ff", "s2t_self", "s2t_stage1", "s2t_stage2", "s2t_ff"]
    seen = 0
    for i, n in enumerate(names):
        key = f"{tag}_v0_sublayer{i}"
        if key not in g.files:
            continue
        ref = g[key]
        got = trace[n].float().cpu().numpy().reshape(ref.shape)
        err = np.abs(got - ref).max()
        assert err <= 1e-3, (tag, n, err)
        seen += 1
    assert seen == 8, seen


@pytest.mark.parametrize("dtype,T", [(torch.float32, 8), (torch.bfloat16, 32), (torch.bfloat16, 64)], ids=["fp32", "bf16_T32", "bf16_T64_permuted"])
def test_lockstep_layer_equals_two_chains_forward_and_gradients(hip, dtype, T):
    """Autograd on, dropout off: the same model, batch and loss through the lock-step form and through the two-chain form.  The two run
    the same kernels on the same rows (batched by direction or not), so the outputs agree to rounding of the few re-ordered sums and
    every parameter gradient agrees closely; bf16 at d_model=512 runs the production kernels (T=64: the region-major t2s form)."""
    from bist_amd import functional as Fn, zbatch as Z
    from bist_amd.model.label_smoothing import LabelSmoothing
    from bist_amd.model.optimize import SimpleLossCompute
    M, Batch = hip
    if dtype == torch.float32:
        cfg, V, C, dims = O.Cfg(d_model=64, att_h=4, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2), 60, 48, (3, T, 9, 7, 11, 8, 6)
    else:
        cfg, V, C, dims = O.Cfg(d_model=512, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2), 300, 256, (4, T, 49, 20, 30, 15, 12)
    B, T_, S, Lq, Lh, Lc, Lt = dims
    ob = O.det_batch(B, T_, S, C, Lq, Lh, Lc, Lt, V, seed=17)
    model, _ = _model(M, cfg, V, C, dtype)
    lc = SimpleLossCompute(model.generator, model.ae_generator, LabelSmoothing(V, O.PAD_ID, 0.1), None, args=_args(cfg))
    res = {}
    old, old_f = Z.ENABLED, Fn.FUSED_TRAIN
    Fn.FUSED_TRAIN = False       # both forms on the SAME stage-1 kernels (the four-launch training forward): what differs is the batching by direction
    try:
        for z in (True, False):
            Z.ENABLED = z
            model.zero_grad(set_to_none=True)
            ft = model.forward(_batch(Batch, ob, dtype))
            terms, logp = lc.terms(ft, _batch(Batch, ob, dtype))
            sum(terms.values()).backward()
            torch.cuda.synchronize()
            res[z] = ({k: ft[k].detach().float().cpu() for k in ("temporal_ft", "spatial_ft", "encoded_ft", "decoded_text")}, logp.detach().float().cpu(),
                      {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        Z.ENABLED, Fn.FUSED_TRAIN = old, old_f
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    for k in res[True][0]:
        assert (res[True][0][k] - res[False][0][k]).abs().max().item() <= tol, k
    assert (res[True][1] - res[False][1]).abs().max().item() <= (1e-4 if dtype == torch.float32 else 6e-2)
    ga, gb = res[True][2], res[False][2]
    assert set(ga) == set(gb) and len(ga) > 100
    bad = {}
    for n in ga:
        a, b = ga[n].flatten().double(), gb[n].flatten().double()
        if n.endswith("linears.1.bias") and b.abs().max().item() < 1e-4:
            continue          # key biases: exactly zero gradient in exact arithmetic (softmax shift invariance), rounding noise in both forms
        scale = max(b.abs().max().item(), 1e-6)
        rel = (a - b).abs().max().item() / scale
        if dtype == torch.float32:
            if rel > 2e-4:
                bad[n] = rel
        else:
            cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item() if b.abs().max() > 0 else 1.0
            if not (cos >= 0.995 and rel <= 8e-2):
                bad[n] = (cos, rel)
    assert not bad, bad


def test_z_linear_dropout_and_layernorm_handoff_against_torch(hip):
    """x [2, M, d] -> y = drop(x . I + 0) + r through ONE z-batched product (identity weights: the output shows the kernel's own dropout
    mask, indexed over the STACKED output) -> LayerNorm of both halves with their own parameters -> a weighted sum.  torch autograd on
    the same arithmetic with that mask gives the reference gradients; the HIP backward takes the masked gradient from the LayerNorm
    backward (`_bist_dz`, mask index offset by the half's first row) instead of a masking pass."""
    from bist_amd import functional as Fn, zbatch as Z
    from bist_amd.model.modules import LayerNorm
    torch.manual_seed(5)
    M_, d, p = 24, 64, 0.3
    dev = "cuda"
    x = torch.randn(2, M_, d, device=dev, requires_grad=True)
    r = torch.randn(2, M_, d, device=dev, requires_grad=True)
    eye = [torch.eye(d, device=dev).requires_grad_(True) for _ in range(2)]
    zb = [torch.zeros(d, device=dev).requires_grad_(True) for _ in range(2)]
    norms = [LayerNorm(d).to(dev) for _ in range(2)]
    for n_ in norms:
        n_.a_2.data.normal_(1.0, 0.2); n_.b_2.data.normal_(0.0, 0.2)
    c1, c2 = torch.randn(2, M_, d, device=dev), torch.randn(2, M_, d, device=dev)
    seed = Fn.next_seed()
    y = Z.linear(x, (eye[0], zb[0]), (eye[1], zb[1]), residual=r, drop_p=p, drop_seed=seed, out_shape=(2, M_, d))
    yn, yr = Z.layernorm_res(y, norms[0], norms[1])
    ((yn * c1).sum() + (yr * c2).sum()).backward()
    torch.cuda.synchronize()
    mask = ((y.detach() - r.detach()).abs() > 0).float()
    keep = mask.mean().item()
    assert abs(keep - (1 - p)) < 0.05 and not torch.equal(mask[0], mask[1]), "the two halves must draw different masks"
    # reference
    xr_, rr_ = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
    a = [n_.a_2.detach().clone().requires_grad_(True) for n_ in norms]
    bb = [n_.b_2.detach().clone().requires_grad_(True) for n_ in norms]
    y_ref = xr_ * mask / (1 - p) + rr_
    yn_ref = torch.stack([O.layer_norm(y_ref[z], a[z], bb[z]) for z in range(2)])
    ((yn_ref * c1).sum() + (y_ref * c2).sum()).backward(retain_graph=True)
    assert (y.detach() - y_ref.detach()).abs().max().item() < 1e-5
    assert (yn.detach() - yn_ref.detach()).abs().max().item() < 1e-4
    for got, ref, what in ((x.grad, xr_.grad, "dx"), (r.grad, rr_.grad, "dr"), (norms[0].a_2.grad, a[0].grad, "da0"), (norms[1].a_2.grad, a[1].grad, "da1"),
                           (norms[0].b_2.grad, bb[0].grad, "db0"), (norms[1].b_2.grad, bb[1].grad, "db1")):
        err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err < 2e-4, (what, err)
    # the weight gradients of the two halves are those of their own rows
    for z in range(2):
        dz_ref = (x.grad[z] * 0 + (mask[z] / (1 - p)))      # d y / d (x.W^T) elementwise factor
        gy = torch.autograd.grad((O.layer_norm(y_ref[z], a[z], bb[z]) * c1[z]).sum() + (y_ref[z] * c2[z]).sum(), y_ref, retain_graph=True)[0][z]
        dw_ref = (gy * dz_ref).t() @ x.detach()[z]
        assert (eye[z].grad - dw_ref).abs().max().item() / max(1.0, dw_ref.abs().max().item()) < 2e-4, z


def test_training_mode_lockstep_step_is_finite_and_changes_with_the_step_counter(hip):
    """train() mode (dropout at all four sites) through the lock-step layer under the Trainer's flat buffers: finite losses, and two
    steps on the same batch draw different dropout masks (the device step counter rides in every z-batched epilogue)."""
    from bist_amd import zbatch as Z
    from bist_amd.train import Trainer
    M, Batch = hip
    cfg = O.Cfg(d_model=128, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2, dropout=0.1)
    V, C = 200, 64
    model, _ = _model(M, cfg, V, C, torch.bfloat16)
    model.train()
    tr = Trainer(model, _args(cfg), V, compute_dtype=torch.bfloat16, warmup=10, use_graph=False)
    ob = O.det_batch(4, 8, 49, C, 20, 30, 12, 10, V, seed=3)
    b = _batch(Batch, ob, torch.bfloat16)
    old = Z.ENABLED
    Z.ENABLED = True
    try:
        l1 = {k: float(v) for k, v in tr.step(b).items()}
        l2 = {k: float(v) for k, v in tr.step(b).items()}
    finally:
        Z.ENABLED = old
    assert all(np.isfinite(v) for v in list(l1.values()) + list(l2.values())), (l1, l2)
    assert l1 != l2
