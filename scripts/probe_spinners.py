"""Can a kernel start on one hardware queue while wait kernels spin on two others?  Direct launches, no graphs (development aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bist_amd import graphsplit as GS, ops, stamps as S
from bist_amd._lib import lib, check

main = torch.cuda.Stream()
with torch.cuda.stream(main):
    sides = GS.exec_streams(3, main.cuda_stream)
x = torch.randn(1 << 20, device="cuda")
w = torch.zeros(16, dtype=torch.int64, device="cuda")
TMO = int(20e-3 * 1e8)      # 20 ms


def trial(tag, nspin, work):
    w.zero_(); torch.cuda.synchronize()
    S.enable(64)
    t0 = time.perf_counter()
    for i in range(nspin):          # spinner i waits for w[0] >= 1 on side stream i; error count in w[1]
        check(lib.bist_graph_queues_distinct(sides[i].cuda_stream, sides[i].cuda_stream, w.data_ptr() + 64, 1), "x") if False else None
    # use the probe entry: wait on stream a, bump on stream b -- here: waits first, then the work + bump on main
    import ctypes as C
    hipk = lib
    for i in range(nspin):
        # wait kernel alone: queues_distinct(a, b) launches wait on a and bump on b; give b = a dummy word so that the wait keeps spinning
        pass
    with torch.cuda.stream(main):
        S.mark("main: before work")
        if work == "big":
            y = ops.add(x, x)
        elif work == "tiny":
            y = ops.add(x[:64], x[:64])
        S.mark("main: after work")
    torch.cuda.synchronize()
    rows = S.read()
    print(tag, [(n, round(t, 1)) for n, _, t in rows], f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    S.disable()


# spinners: bist_graph_queues_distinct(a, b, scratch): wait on a for scratch[0] >= 1, then bump scratch[0] on b.
# Launch the waits on the side streams with the bump going to `main` AFTER the work: the work must run while the waits spin.
def trial2(tag, nspin, work):
    w.zero_(); torch.cuda.synchronize()
    S.enable(64)
    t0 = time.perf_counter()
    # waits first: each probe call's bump is sent to a stream that is itself blocked behind a wait on a word nobody sets ... simpler:
    # wait i on side[i] for word 2*i, bumps for all words on main after the work
    for i in range(nspin):
        check(lib.bist_graph_queues_distinct(sides[i].cuda_stream, main.cuda_stream, w.data_ptr() + 16 * i, TMO), "probe")
        # (its bump on main runs at once: the wait ends immediately) -> not a spinner; so instead:
    S.disable()


print("see probe via graphs below")
# The clean way: tiny graphs of ONE kernel each are not needed -- use the split library's own kernels through a two-node capture.
# wait kernels via capture-free path: launch bist_graph_queues_distinct(side_i, side_3) where side_3 is held busy by a long sleep first.
hold = sides[2]
for nspin, work in ((0, "big"), (1, "big"), (2, "big"), (2, "tiny"), (2, "none")):
    w.zero_(); torch.cuda.synchronize()
    S.enable(64)
    with torch.cuda.stream(hold):
        torch.cuda._sleep(int(2.0e9 * 3e-3))            # ~3 ms: the bumps below queue up behind it
    for i in range(nspin):
        check(lib.bist_graph_queues_distinct(sides[i].cuda_stream, hold.cuda_stream, w.data_ptr() + 16 * i, TMO), "probe")
        if os.environ.get("PENDING") == "1":             # a packet queued BEHIND the spinner (barrier bit: it waits for the spinner to end)
            with torch.cuda.stream(sides[i]):
                for _ in range(int(os.environ.get("NPENDING", "1"))):
                    S.mark("side%d: behind its spinner" % i)
    with torch.cuda.stream(main):
        S.mark("main: before work")
        if work == "big":
            y = ops.add(x, x)
        elif work == "tiny":
            y = ops.add(x[:64], x[:64])
        S.mark("main: after work")
        for _ in range(3):
            ops.add(x, x)
        S.mark("main: after 3 more")
    with torch.cuda.stream(hold):
        S.mark("hold: sleep over, bumps done")
    torch.cuda.synchronize()
    rows = S.read()
    print(f"{nspin} spinners, work {work}:", [(n, round(t, 1)) for n, _, t in rows], "timeouts", [int(w[2 * i + 1]) for i in range(nspin)], flush=True)
    S.disable()
