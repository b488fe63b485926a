"""Does a replayed hipGraph run captured branches side by side?  Two / three independent chains of small launches captured on forked
streams against the same launches on one stream (development aid)."""
import time
import torch

dev = "cuda"
N = 40            # launches per chain
xs = [torch.randn(320, 512, device=dev, dtype=torch.bfloat16) for _ in range(3)]
ws = [torch.randn(512, 512, device=dev, dtype=torch.bfloat16) * 0.04 for _ in range(3)]


def chain(i):
    y = xs[i]
    for _ in range(N):
        y = torch.tanh(y @ ws[i])
    return y


def capture(nbranch, forked):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(nbranch - 1)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(nbranch):
            chain(i)                                        # warm-up outside capture
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            main = torch.cuda.current_stream()
            outs = []
            if forked:
                for i, st in enumerate(side):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        outs.append(chain(i + 1))
                outs.append(chain(0))
                for st in side:
                    main.wait_stream(st)
            else:
                for i in range(nbranch):
                    outs.append(chain(i))
    return g, outs


def timeit(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for nb in (1, 2, 3):
    for forked in ((False,) if nb == 1 else (False, True)):
        g, o = capture(nb, forked)
        print(f"{nb} chain(s) of {2 * N} launches, {'forked streams' if forked else 'one stream'}: {timeit(g):.3f} ms per replay", flush=True)
