import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
import bist_amd.model as M
from bist_amd import ops
from bist_amd.data.synthetic import synthetic_batch
from bist_amd.train import Trainer
import bist_amd.train as T
c = dict(bench.CFG)
args = bench.model_args(c["L"], c["d"], c["h"], 0.1)
torch.manual_seed(1)
model = M.make_model(c["V"], c["V"], args, ft_sizes=[c["C"]]).cuda(); model.train()
tr = Trainer(model, args, c["V"], compute_dtype=torch.bfloat16, use_graph=True)
b = synthetic_batch(c["B"], T=c["T"], S=c["S"], C=c["C"], Lq=c["Lq"], Lh=c["Lh"], Lc=c["Lc"], Lt=c["Lt"], vocab=c["V"], seed=1234, dtype=torch.bfloat16)
orig_cs, orig_ln = ops.col_sum_flush, ops.lngrad_flush
def cs():
    q = ops.COLSUM_QUEUE or []
    print("col_sum flush: %d jobs, rows %s" % (len(q), sorted({j[2] for j in q})), "MB", round(sum(j[2]*j[3]*2 for j in q)/1e6,1), "capturing", torch.cuda.is_current_stream_capturing())
    orig_cs()
def ln():
    q = ops.LNGRAD_QUEUE or []
    print("lngrad flush: %d jobs, rows %s" % (len(q), sorted({j[1].shape[0] for j in q})), "MB", round(sum(j[1].numel()*4 for j in q)/1e6,1))
    orig_ln()
ops.col_sum_flush, ops.lngrad_flush = cs, ln
T.ops.col_sum_flush, T.ops.lngrad_flush = cs, ln
tr.step(b); torch.cuda.synchronize()
