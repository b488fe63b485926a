"""CPU tests of the N>1 host logic with the gloo backend, world_size 2 (the GPU path uses the same code with RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bist_amd import parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    n = 10_000
    local = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank-specific "gradient"
    for bucket in (0, 3000):
        g = local.clone()
        scale = parallel.exchange_gradients(g, bucket_elems=bucket)
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert scale == 1.0 / world and torch.equal(g, want), (bucket, rank)
    # the trainer's form: pieces issued back to back, each consumed as soon as its own all-reduce is done
    g = local.clone()
    bounds = parallel.chunk_bounds(n, 8, 64)
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert all(lo % 64 == 0 for lo, _ in bounds) and len(bounds) <= 8
    seen = torch.zeros(n)
    for (lo, hi), wk in zip(bounds, parallel.exchange_gradients_async(g, bounds)):
        wk.wait()
        seen[lo:hi] = g[lo:hi]                                           # what an optimiser launched here would read
    assert torch.equal(seen, want), rank
    # one SGD-like update with the exchanged gradient is identical on every rank
    w = torch.ones(n) - 0.1 * scale * g
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, w)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    out.put((rank, parallel.clip_range(rank, world, 33)))
    dist.destroy_process_group()


def test_gradient_exchange_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ranges = dict(q.get() for _ in range(world))
    assert ranges[0] == (0, 17) and ranges[1] == (17, 33)            # contiguous, covering, sizes differ by <= 1


def test_single_process_is_identity():
    g = torch.ones(5)
    assert parallel.exchange_gradients(g) == 1.0 and torch.equal(g, torch.ones(5))
    assert abs(parallel.noam_rate(1, 512) - 512 ** -0.5 * 4000 ** -1.5) < 1e-12
    assert parallel.noam_rate(4000, 512) > parallel.noam_rate(8000, 512) > 0
    assert parallel.chunk_bounds(10, 3) == [(0, 4), (4, 8), (8, 10)] and parallel.chunk_bounds(5, 8, 64) == [(0, 5)]
