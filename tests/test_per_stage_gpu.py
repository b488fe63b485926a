"""Per-stage parity of the reasoning layer: every sublayer output of VidEncoderLayer4 against the reference's own golden tensors."""
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
def test_per_stage_outputs_match_the_reference_golden(hip, golden_dir, tag):
    """VidEncoderLayer4's sublayer outputs A0, A1, A2, F0, A3, A4, A5, F1 of layer 0 (encoder.py:121,130,135,148,161,166 through the
    reference's own SublayerConnection hooks, tests/golden/g3_model.npz `*_v0_sublayer{i}`) against the HIP path's per-stage tensors
    (debug hook `_bist_trace`), fp32, 1e-3."""
    M, Batch = hip
    g = np.load(os.path.join(golden_dir, "g3_model.npz"))
    meta = json.loads(str(g[f"{tag}_cfg"]))
    cfg, dm = O.Cfg(**meta["cfg"]), meta["dims"]
    ob = O.det_batch(dm["B"], dm["T"], dm["S"], dm["C"], dm["Lq"], dm["Lh"], dm["Lc"], dm["Lt"], dm["V"], seed=dm.get("seed", 1234),
                     fully_masked_clip=meta.get("fully_masked", False))
    model, _ = _model(M, cfg, dm["V"], dm["C"], torch.float32)
    layer = model.mutlimodal_decoder.v_layers[0]
    trace = layer.__dict__["_bist_trace"] = {}
    try:
        with torch.no_grad():
            model.forward(_batch(Batch, ob, torch.float32))
    finally:
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
