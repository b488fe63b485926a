"""How does the HIP runtime order the branches of a multi-stream hipGraph?  (development aid, round 4)

A captured graph of three streams: `main` forks `side` (a LONG kernel), goes on with short work, then forks `side2` whose first node
depends on main's short work only.  If side2's node starts right after main's short work the replay honours the captured dependencies;
if it starts after the long kernel the runtime has put it on the long kernel's queue (or waits for that queue's tail).
Timestamps: bist_dev_timestamp launches (100 MHz device clock).
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import stamps as S, functional as Fn

dev = torch.device("cuda")
main_work = torch.zeros(1 << 16, device=dev)
LONG = int(2.0e9 * 300e-6)       # ~300 us


def variant(order):
    """order: 'long_first' issues the long side kernel before main's short work (capture order), 'long_last' after it."""
    S.enable(256)
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    warm = torch.cuda.Stream()
    with torch.cuda.stream(warm):
        torch.cuda._sleep(1000); main_work.add_(1)
    torch.cuda.synchronize()
    S.NAMES.clear()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream()
        S.mark("main head")
        main_work.add_(1)
        ev0 = torch.cuda.Event(); ev0.record(main)

        def long_branch():
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                S.mark("side: long begins")
                torch.cuda._sleep(LONG)
                S.mark("side: long ended")

        def short_then_fork():
            for _ in range(4):
                main_work.add_(1)
            S.mark("main: short work done")
            side2.wait_stream(main)
            with torch.cuda.stream(side2):
                S.mark("side2: first node (depends on main's short work only)")
                main_work.add_(1)
                S.mark("side2: done")
        if order == "long_first":
            long_branch(); short_then_fork()
        else:
            short_then_fork(); long_branch()
        main.wait_stream(side); main.wait_stream(side2)
        S.mark("main: joined")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    S.BUF.zero_(); torch.cuda.synchronize()
    g.replay()
    rows = S.read()
    print(f"--- capture order: {order}; DEBUG_HIP_FORCE_GRAPH_QUEUES={os.environ.get('DEBUG_HIP_FORCE_GRAPH_QUEUES')}")
    for n, s, t in rows:
        print("%8.1f us  %s" % (t, n))
    S.disable()


variant("long_first")
variant("long_last")
