import torch
a = torch.randn(4096, 4096, device="cuda")
b = torch.randn(4096, 4096, device="cuda")
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print("no external events:", ex); raise SystemExit
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    c = a @ b
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    x = a + 1
    e0.record()
    c = a @ b
    e1.record()
    y = c + 1
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "elapsed ms", e0.elapsed_time(e1))
