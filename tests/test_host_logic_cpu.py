"""Host-side logic of the decode path that needs no GPU: the parameter walk behind the graph store's staleness key and the length
buckets of a dialogue's token tensors (the reference pads nothing: data/dataset.py builds exact-length batches)."""
import copy
import types

import torch
import torch.nn as nn

from bist_amd import ops


def _net():
    return nn.Sequential(nn.Linear(4, 4), nn.Sequential(nn.LayerNorm(4), nn.Linear(4, 2)))


def test_module_parameters_are_the_current_parameter_objects():
    """ops.module_parameters (the decode turn's cheap stand-in for nn.Module.parameters()) returns the Parameter objects as they are now:
    a REPLACED Parameter, an in-place update (version counter) and an ADDED submodule are all seen."""
    net = _net()
    assert set(map(id, ops.module_parameters(net))) == set(map(id, net.parameters()))
    v0 = sum(p._version for p in ops.module_parameters(net))
    with torch.no_grad():
        net[0].weight.add_(1.0)
    assert sum(p._version for p in ops.module_parameters(net)) == v0 + 1
    new = nn.Parameter(torch.zeros(4, 4))
    net[0].weight = new                                   # replaced object: the cached module list still reads the modules' dicts
    assert any(p is new for p in ops.module_parameters(net))
    net[1].add_module("extra", nn.Linear(2, 2))           # a module gained a child: the cached list is rebuilt
    assert set(map(id, ops.module_parameters(net))) == set(map(id, net.parameters()))
    del net[1][0]                                         # ... or lost one
    assert set(map(id, ops.module_parameters(net))) == set(map(id, net.parameters()))


def test_length_buckets_pad_tokens_and_masks_only():
    """decode._bucketed: query / history / caption token tensors padded with the pad id to multiples of the bucket, their masks with False
    (so every attention and pointer head excludes the new positions), the stacked query mask rebuilt, nothing else touched; lengths that
    already are multiples stay the same objects."""
    import bist_amd.model.decode as D
    pad = 1
    b = types.SimpleNamespace(
        query=torch.randint(2, 9, (1, 5)), his=torch.randint(2, 9, (1, 16)), cap=torch.randint(2, 9, (1, 9)),
        query_mask=torch.ones(1, 1, 5, dtype=torch.bool), his_mask=torch.ones(1, 1, 16, dtype=torch.bool), cap_mask=torch.ones(1, 1, 9, dtype=torch.bool),
        fts=torch.zeros(1, 3, 4))
    b.query_mask2 = torch.cat([b.query_mask, b.query_mask], dim=0)
    old = D.BUCKET
    try:
        D.BUCKET = 8
        out = D._bucketed(copy.copy(b), pad)
        assert out.query.shape == (1, 8) and out.cap.shape == (1, 16) and out.his is b.his and out.his_mask is b.his_mask
        assert torch.equal(out.query[:, :5], b.query) and (out.query[:, 5:] == pad).all()
        assert out.query_mask.shape == (1, 1, 8) and out.query_mask[..., :5].all() and not out.query_mask[..., 5:].any()
        assert out.cap_mask.shape == (1, 1, 16) and not out.cap_mask[..., 9:].any()
        assert out.query_mask2.shape == (2, 1, 8) and torch.equal(out.query_mask2[0], out.query_mask[0])
        assert out.fts is b.fts
        D.BUCKET = 0
        assert D._bucketed(b, pad) is b
    finally:
        D.BUCKET = old
