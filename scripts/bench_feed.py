"""PCIe-inclusive training rate (development aid): the bench's training step with the features
handed over as host fp32 tensors each step through bist_amd.data.feeder.DeviceFeeder, against features resident in HBM."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bist_amd.model as M
from bist_amd import functional as Fn
from bist_amd.data.feeder import DeviceFeeder, HostBatch
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer

c = bench.CFG
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
Fn.manual_seed(1234)
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
dev_b = synthetic_batch(c["B"], dtype=torch.bfloat16, seed=1234)
steps = 20
for _ in range(3): tr.step(dev_b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): tr.step(dev_b)
torch.cuda.synchronize(); resident = (time.perf_counter() - t0) / steps
for label, pinned in (("pageable fp32 producer", False), ("pinned fp32 producer", True)):
    host = []
    for i in range(2):
        b = synthetic_batch(c["B"], dtype=torch.float32, seed=1234 + i, device="cpu")
        f = b.fts
        if pinned:
            p = DeviceFeeder.pinned_like(f.shape, f.dtype); p.copy_(f); f = p
        host.append(HostBatch(b.query, b.his, f, b.cap, b.trg, b.trg_y))
    src = [host[i % 2] for i in range(steps + 3)]
    it = iter(DeviceFeeder(src, feature_dtype=torch.bfloat16))
    for _ in range(3): tr.step(next(it))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for b in it:
        tr.step(b); n += 1
    torch.cuda.synchronize(); fed = (time.perf_counter() - t0) / n
    print(f"{label}: {fed*1e3:.2f} ms/step ({int(dev_b.ntokens)/fed:.0f} tok/s) vs resident {resident*1e3:.2f} ms/step ({int(dev_b.ntokens)/resident:.0f} tok/s); "
          f"features {host[0].fts.numel()*4/1e6:.0f} MB fp32 per step")
