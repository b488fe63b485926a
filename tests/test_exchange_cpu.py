"""Host logic of the overlapped gradient exchange (train.py: buckets of the last layers' matrices exchanged DURING the backward pass)."""
import argparse
import os
import socket

import torch
import torch.distributed as dist


def test_bucket_mark_fires_after_everything_recorded_behind_it():
    """Fn.bucket_mark relies on the autograd engine running nodes in decreasing sequence number: the mark's backward must come after the
    backward of EVERY node recorded later in the forward pass -- also of nodes that do not depend on the marked tensor (the ahead-issued
    value projections, the other streams' layers) -- and before that of the nodes recorded earlier."""
    from bist_amd import functional as Fn
    order = []

    def logged(t, name):
        t.register_hook(lambda g: order.append(name))
        return t
    a = torch.randn(4, requires_grad=True)
    w = torch.randn(4, requires_grad=True)
    early = logged(a * 2.0, "early")                                # recorded BEFORE the mark
    Fn.BUCKET_MARK = ({3}, lambda k: order.append("mark%d" % k))
    try:
        assert Fn.bucket_mark(early, 2) is early                    # not a bucket boundary: identity, no node
        m = Fn.bucket_mark(early, 3)
    finally:
        Fn.BUCKET_MARK = None
    independent = logged(w * 3.0, "independent")                    # behind the mark, not downstream of it
    late = logged(m + 1.0, "late")
    later = logged(late * independent, "later")
    later.sum().backward()
    assert order.index("mark3") > max(order.index(n) for n in ("later", "late", "independent")), order
    assert order.index("mark3") < order.index("early"), order
    with torch.no_grad():
        Fn.BUCKET_MARK = ({3}, lambda k: order.append("never"))
        try:
            assert Fn.bucket_mark(early, 3) is early                # inference: identity
        finally:
            Fn.BUCKET_MARK = None


def test_bucket_layout_covers_the_last_layers_in_the_order_they_become_final():
    import bist_amd.model as M
    from bist_amd import train as T
    from oracle import bist_oracle as O
    cfg = O.Cfg(d_model=64, att_h=4, nb_blocks=6, nb_venc_blocks=6, nb_cenc_blocks=6)
    args = argparse.Namespace(**{**cfg.__dict__, "d_ff": 4 * cfg.d_model})
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")
    os.environ["BIST_FORCE_EXCHANGE"] = "1"
    try:
        torch.manual_seed(0)
        model = M.make_model(80, 80, args, ft_sizes=[64])
        t = T.Trainer(model, args, 80, compute_dtype=torch.float32, use_graph=False)
        assert t.exchanging and not t.overlap                       # CPU tensors: no streams to overlap on, the layout is the same
        names = {id(p): n for n, p in model.named_parameters()}
        offs = {names[id(p)]: p._grad_view.data_ptr() - t.flat_grad.data_ptr() for p in t.params}
        assert [c for c, _, _ in t.buckets] == [4, 2] == sorted((c for c, _, _ in t.buckets), reverse=True)
        assert t.buckets[0][2] == t.numel and t.buckets[1][2] == t.buckets[0][1] and t._bucket_end == t.buckets[1][1] > t.n32
        es = t.flat_grad.element_size()
        for name, off in offs.items():
            if ".weight" not in name or "norm" in name or "lut" in name:
                continue
            import re
            mt = re.match(r"mutlimodal_decoder\.(?:v_layers|c_layers|layers)\.(\d+)\.", name)
            layer = int(mt.group(1)) if mt else -1
            el = off // es
            where = next((j for j, (_, lo, hi) in enumerate(t.buckets) if lo <= el < hi), None)
            want = 0 if layer >= 4 else 1 if layer >= 2 else None
            if el >= t.n32:                                          # (the matrices; biases / LayerNorm parameters live in the prefix)
                assert where == want, (name, layer, where)
    finally:
        os.environ.pop("BIST_FORCE_EXCHANGE", None)
        dist.destroy_process_group()
