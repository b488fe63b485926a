"""Host-to-device bandwidth of the feature tensor of one training batch (205 MB fp32, pinned): one copy on one stream, two / four pieces on as
many streams (development aid, round 4: is the PCIe-inclusive step bound by one SDMA engine or by the link?)."""
import time, torch
n = 16 * 32 * 49 * 2048
h = torch.empty(n, dtype=torch.float32).pin_memory(); h.normal_()
d = torch.empty(n, dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
for parts in (1, 2, 4, 1, 2, 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        step = n // parts
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                d[p * step:(p + 1) * step].copy_(h[p * step:(p + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{parts} piece(s): {dt * 1e3:.2f} ms per 205 MB = {n * 4 / dt / 1e9:.1f} GB/s", flush=True)
