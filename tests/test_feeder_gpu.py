"""Input side: DeviceFeeder hands over the same device Batch as a direct move_to_cuda, for fp32 and
bf16 features, pinned and pageable producers, and the transfer of the next batch does not disturb the current one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_batches(n, B=3, T=6, S=5, C=64, pinned=False, seed=0):
    from bist_amd.data.feeder import DeviceFeeder, HostBatch
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        fts = torch.from_numpy(rs.standard_normal((B, T, S, C)).astype(np.float32))
        fts[i % B, T // 2:] = 0.0                                   # zero temporal rows drive temporal_mask (dataset.py:79)
        ids = lambda L: torch.from_numpy(rs.randint(1, 50, size=(B, L)).astype(np.int64))
        q, h, c, t, ty = ids(7), ids(11), ids(5), ids(6), ids(6)
        if pinned:
            p = DeviceFeeder.pinned_like(fts.shape, fts.dtype); p.copy_(fts); fts = p
        out.append(HostBatch(q, h, fts, c, t, ty))
    return out


@pytest.mark.parametrize("feature_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pinned", [False, True])
def test_feeder_matches_direct_batches(feature_dtype, pinned):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from bist_amd.data.batch import Batch
    from bist_amd.data.feeder import DeviceFeeder
    hbs = _host_batches(5, pinned=pinned)
    kept = []
    for i, b in enumerate(DeviceFeeder(hbs, feature_dtype=feature_dtype)):
        hb = hbs[i]
        ref = Batch(hb.query.cuda(), hb.his.cuda(), hb.fts.cuda().to(feature_dtype), hb.cap.cuda(), hb.trg.cuda(), hb.trg_y.cuda())
        assert b.fts.dtype == feature_dtype and torch.equal(b.fts, ref.fts)
        for name in ("query", "his", "cap", "trg", "trg_y", "query_mask", "his_mask", "cap_mask", "trg_mask", "temporal_mask", "trg_mean_mask"):
            assert torch.equal(getattr(b, name), getattr(ref, name)), name
        assert int(b.ntokens) == int(ref.ntokens)
        # a host-side temporal mask computed the reference's way (dataset.py:79) agrees with the device one
        assert torch.equal(b.temporal_mask.cpu(), (hb.fts.sum(2).sum(-1) != 0).unsqueeze(-2))
        kept.append(i)
        torch.cuda.synchronize()                                    # batch i+1 is already crossing PCIe into the other slot:
        assert torch.equal(b.fts, ref.fts)                          # the current batch is untouched by it
    assert len(kept) == 5


def test_feeder_host_running_ahead_of_a_slow_consumer():
    """No per-iteration synchronise and a consumer that is slow on the device: the host runs several batches ahead, so the
    pinned staging buffers of batch i-2 are rewritten while its H2D copy may still be queued -- the feeder must hold the host
    until that copy has read them (pageable producers, big features so the copies take a while)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from bist_amd.data.feeder import DeviceFeeder
    hbs = _host_batches(8, B=4, T=16, S=49, C=512, seed=3)
    sums = []
    for b in DeviceFeeder(hbs, feature_dtype=torch.float32):
        torch.cuda._sleep(40_000_000)                               # ~20 ms of device time in front of the consumer's read
        sums.append(b.fts.double().sum())                           # stays on the device: nothing here waits for the GPU
    torch.cuda.synchronize()
    for i, s in enumerate(sums):
        assert abs(float(s) - float(hbs[i].fts.double().sum())) < 1e-6, i
