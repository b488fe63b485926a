import torch, time
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
h = torch.empty(206*1024*1024//4, dtype=torch.float32, pin_memory=True)
d = torch.empty_like(h, device="cuda")
cs = torch.cuda.Stream()
def compute():
    for _ in range(20): y = x @ x
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter()-t0)*1e3
compute(); 
print("compute", t(compute))
def copy():
    with torch.cuda.stream(cs): d.copy_(h, non_blocking=True)
copy(); print("copy", t(copy))
def both():
    copy(); compute()
print("both", t(both))
