"""The drop-in boundary of INTEGRATION.md section 2, executed: with the reference's package name ``model`` mapped onto this build,
the import lines of train.py:16-18 and the calls of train.py:91-135 / generate.py:93 run unchanged --

  * a whole-module pickle WRITTEN BY THE REFERENCE (tests/golden/g7_reference_module.pth.tar, train.py:161) is unpickled by its
    ``model.*`` class paths into this build's classes and reproduces the reference's outputs;
  * ``make_model`` + ``SimpleLossCompute(opt=NoamOpt(model_size, 1, warmup, torch.optim.Adam(...)))`` + a ``run_epoch``-shaped loop
    (train.py:21-52) at BASELINE configs[0] dims (L=2, d_model=128, B=32) follow the CPU oracle's losses step for step;
  * ``torch.save(model)`` / ``torch.load`` round-trips the trained module.
"""
import argparse
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import bist_oracle as O

pytestmark = pytest.mark.gpu
NAMES = ("", ".mtn", ".modules", ".encoder", ".decoder", ".generator", ".label_smoothing", ".optimize", ".decode")


@pytest.fixture()
def swapped():
    """INTEGRATION.md section 2, verbatim; undone afterwards."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    import bist_amd.model as m
    saved = {("model" + n): sys.modules.get("model" + n) for n in NAMES}
    for name in NAMES:
        sys.modules["model" + name] = sys.modules["bist_amd.model" + name] if name else m
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _err(a, b):
    return (a.detach().float().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def test_reference_whole_module_pickle_loads_and_matches(swapped, golden_dir):
    from bist_amd.data.batch import Batch
    import bist_amd.model.mtn as mtn
    g = np.load(os.path.join(golden_dir, "g7_reference_module.npz"))
    meta = json.loads(str(g["cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    model = torch.load(os.path.join(golden_dir, "g7_reference_module.pth.tar"), weights_only=False)      # generate.py:93
    assert isinstance(model, mtn.MTN)
    model = model.cuda().eval()
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm["seed"])
    b = Batch(ob.query.cuda(), ob.his.cuda(), ob.fts.cuda(), ob.cap.cuda(), ob.trg.cuda(), ob.trg_y.cuda())
    with torch.no_grad():
        ft = model.forward(b)
        logp = model.generator(ft, b, model.args)
    for k in [k for k in g.files if k.startswith("ft_")]:
        assert _err(ft[k[3:]], g[k]) <= 1e-3, k
    assert _err(logp, g["logp"]) <= 1e-3
    assert np.array_equal(logp.argmax(-1).cpu().numpy(), g["logp"].argmax(-1))


def test_train_script_calls_follow_the_oracle_step_for_step(swapped):
    ns = {}
    exec("from model.mtn import *\nfrom model.label_smoothing import *\nfrom model.optimize import *", ns)      # train.py:16-18
    from bist_amd.data.batch import Batch
    from bist_amd.train import run_epoch
    cfg = O.Cfg(d_model=128, att_h=8, nb_blocks=2, nb_venc_blocks=2, nb_cenc_blocks=2)          # BASELINE configs[0]
    V, C, B, T, S = 500, 2048, 32, 32, 49
    args = argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model, "warmup_steps": 40, "num_epochs": 1, "report_interval": 1})
    sd = O.det_state(cfg, V, C)
    model = ns["make_model"](V, V, args, ft_sizes=[C])                                                           # train.py:91
    model.load_state_dict(sd, strict=False)
    model.cuda()                                                                                                 # train.py:92
    criterion = ns["LabelSmoothing"](size=V, padding_idx=O.PAD_ID, smoothing=0.1)                                # train.py:93
    model_opt = ns["NoamOpt"](args.d_model, 1, args.warmup_steps, torch.optim.Adam(model.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
    train_loss = ns["SimpleLossCompute"](model.generator, model.ae_generator, criterion, opt=model_opt, l=1.0, args=args)   # train.py:129-135
    obs = [O.det_batch(B, T, S, C, 20, 30, 12, 10, V, seed=40 + i) for i in range(3)]
    loader = [Batch(o.query.cuda(), o.his.cuda(), o.fts.cuda(), o.cap.cuda(), o.trg.cuda(), o.trg_y.cuda()) for o in obs]
    seen = []
    model.eval()            # dropout off, as in the oracle (the optimiser still steps: SimpleLossCompute owns backward + step)
    out_losses = run_epoch(None, loader, None, 0, model, train_loss, report=lambda j, l, b: seen.append({k: float(v) for k, v in l.items()}))
    # the oracle: same weights, same batches, torch Adam with the Noam rate (optimize.py:19-34)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj")}
    for a in ("tgt_embed.0.lut.weight", "generator.vocab_gen", "ae_generator.proj"):
        leaf[a] = leaf["query_embed.0.lut.weight"]
    params = list({id(v): v for v in leaf.values()}.values())
    opt = torch.optim.Adam(params, lr=0, betas=(0.9, 0.98), eps=1e-9)
    for step, ob in enumerate(obs, 1):
        ft = O.mtn_forward(leaf, cfg, ob)
        l = O.loss_compute(leaf, cfg, ft, ob, V)
        l["total"].backward()
        for gp in opt.param_groups:
            gp["lr"] = args.d_model ** -0.5 * min(step ** -0.5, step * args.warmup_steps ** -1.5)
        opt.step(); opt.zero_grad(set_to_none=True)
        want = {"out": float(l["out"].detach() * ob.ntokens), "temporal_ae": float(l["temporal_ae"].detach() * ob.qntokens),
                "spatial_ae": float(l["spatial_ae"].detach() * ob.qntokens)}
        for k, v in want.items():          # the reference reports the un-normalised sums (optimize.py:89-93)
            assert abs(seen[step - 1][k] - v) <= 2e-3 * step * abs(v), (step, k, seen[step - 1][k], v)
    assert set(out_losses) == {"out", "temporal_ae", "spatial_ae"} and all(torch.isfinite(v) for v in out_losses.values())
    # train.py:161 / generate.py:93: whole-module save and load
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    with torch.no_grad():
        a1, a2 = model.forward(loader[0]), again.forward(loader[0])
    assert torch.equal(a1["decoded_text"], a2["decoded_text"])


def test_generic_sublayer_and_position_modules_in_training_mode(swapped):
    """The reference's generic call forms with dropout ON (modules.py:42-44, 142-144): outputs equal x + mask*f(LN(x))/(1-p) for a
    0/1 mask with the configured rate, and the backward uses that same mask."""
    from bist_amd.model.modules import PositionalEncoding, SublayerConnection
    from bist_amd import functional as Fn
    Fn.manual_seed(7)
    d, p = 64, 0.3
    sub = SublayerConnection(d, p).cuda().train()
    x = torch.randn(5, 11, d, device="cuda", requires_grad=True)
    w = torch.randn(5, 11, d, device="cuda")
    y = sub(x, lambda t: t * 2.0 + 1.0)
    ln = sub.norm(x.detach())
    f = ln * 2.0 + 1.0
    m = (y.detach() - x.detach()) / f * (1 - p)                    # recovered mask
    assert ((m - m.round()).abs().max().item() < 1e-3) and abs(1 - m.round().mean().item() - p) < 0.05
    (y * w).sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    pe = PositionalEncoding(d, p).cuda().train()
    z = torch.randn(3, 9, d, device="cuda")
    out = pe(z)
    full = z + pe.pe[0, :9].to(z.dtype)
    mm = out / full * (1 - p)
    assert (mm - mm.round()).abs().max().item() < 1e-3 and abs(1 - mm.round().mean().item() - p) < 0.06
