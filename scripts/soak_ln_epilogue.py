"""200 launches of the input projection with the LayerNorm epilogue at B = 64 (784 workgroup pairs shaking hands each): results bit-identical, header words back to zero (development aid)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import ops
ops.LN_EPILOGUE = True
torch.manual_seed(0)
M, K = 100352, 2048
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(512, K, device="cuda") * (1.0 / K ** 0.5)).bfloat16()
bias = (torch.randn(512, device="cuda") * 0.1).bfloat16()
a = (1 + 0.2 * torch.randn(512, device="cuda")).bfloat16(); b = (0.1 * torch.randn(512, device="cuda")).bfloat16()
first = ops.linear(x, w, bias, act=ops.ACT_RELU, ln_out=(a, b, 1e-6)).clone()
bad = 0
for it in range(200):
    y = ops.linear(x, w, bias, act=ops.ACT_RELU, ln_out=(a, b, 1e-6))
    if not torch.equal(y, first): bad += 1
torch.cuda.synchronize()
ws = ops._workspace(x.device)
print("mismatches", bad, "header max", int(ws.view(torch.int32)[:1024].abs().max().item()))
