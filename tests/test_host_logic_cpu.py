"""Host-side logic of the decode path that needs no GPU: the parameter walk behind the graph store's staleness key and the length
buckets of a dialogue's token tensors (the reference pads nothing: data/dataset.py builds exact-length batches)."""
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
    net[0] = nn.Linear(4, 4)                              # ... or had one replaced (same count)
    assert set(map(id, ops.module_parameters(net))) == set(map(id, net.parameters()))


def test_length_buckets_widen_tokens_and_masks_only():
    """decode._staged_shape: the static buffers of a turn's graphs hold query / history / caption tokens and their masks at the next multiple
    of the length bucket (bist_stage_inputs fills the tail with the pad id / False; tests/test_ops_gpu.py), every other field at its own
    shape; bucket 0 / 1 = exact shapes."""
    import bist_amd.model.decode as D
    fields = {"query": (1, 5), "his": (1, 16), "cap": (1, 9), "query_mask": (1, 1, 5), "query_mask2": (2, 1, 5), "his_mask": (1, 1, 16),
              "cap_mask": (1, 1, 9), "fts": (1, 3, 4), "temporal_mask": (1, 1, 3)}
    want8 = dict(fields, query=(1, 8), cap=(1, 16), query_mask=(1, 1, 8), query_mask2=(2, 1, 8), cap_mask=(1, 1, 16))
    for f, shp in fields.items():
        v = torch.zeros(shp)
        assert D._staged_shape(f, v, 8) == want8[f], f
        assert D._staged_shape(f, v, 0) == shp and D._staged_shape(f, v, 1) == shp
    assert set(D._BUCKETED_FIELDS) <= set(D._TURN_FIELDS)
    # "class": the padded lengths of the decoder kernel's caches
    for f, L, P in (("query", 5, 32), ("query", 32, 32), ("query", 33, 64), ("his", 2, 32), ("his", 33, 64), ("his", 64, 64), ("his", 65, 128),
                    ("his", 129, 256), ("his", 256, 256), ("his", 257, 512), ("his", 512, 512), ("his", 513, 576), ("cap", 9, 16), ("cap_mask", 25, 32), ("his_mask", 70, 128),
                    ("query_mask2", 20, 32)):
        assert D._staged_shape(f, torch.zeros(1, 1, L), "class") == (1, 1, P), (f, L)
    assert D._staged_shape("fts", torch.zeros(1, 3, 4), "class") == (1, 3, 4)
