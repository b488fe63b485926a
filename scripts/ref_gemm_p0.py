"""What the vendor library reaches on the P0 shape (development aid): torch.nn.functional.linear in bf16 at [B*T*S, 2048] x [2048 -> 512]."""
import sys, torch
for B in (16, 64):
    M, K, N = B * 32 * 49, 2048, 512
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = torch.nn.functional.linear(x, w, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B}: M={M}: {ms * 1e3:.1f} us per product, {2 * M * K * N / ms / 1e9:.0f} TFLOP/s")
