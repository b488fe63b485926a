"""Phase timeline of decstack_kernel (workgroup 0, layer 0, self-attention sublayer) from s_memtime stamps (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stamps = torch.zeros(128, dtype=torch.int64, device="cuda")
from bist_amd import _lib as _bl
_bl.check(_bl.lib.bist_dev_set_stamps(1, stamps.data_ptr()), "bist_dev_set_stamps")      # explicit hand-over of a buffer this script owns
import bench
import bist_amd.model as M
from bist_amd.model.decode import beam_search_decode
from bist_amd.data.synthetic import synthetic_batch
c = bench.CFG
args = bench.model_args(6, 512, 8, 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda().to(torch.bfloat16).eval()
b1 = synthetic_batch(1, T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=99, dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(2):
        beam_search_decode(model, b1, 12, 2, 0, 3, 1, beam=5, penalty=1.0, nbest=5, train_args=args)
torch.cuda.synchronize()
st = [v for v in stamps.cpu().tolist() if v]
d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
print("ticks between grid-barrier exits, layer 0:", d[:14])
print("layer 3:", d[42:56])
print("total", st[-1] - st[0], "over", len(d), "phases")
