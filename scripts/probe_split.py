"""Debug aid for the split-graph executor: the three-stream test DAG under launch-order / main-stream variants."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bist_amd import graphsplit as GS, ops
from bist_amd._lib import lib, check
import ctypes as C
from test_graphsplit_gpu import _issue

x = torch.randn(1 << 14, device="cuda") * 0.01
streams = (torch.cuda.Stream(), torch.cuda.Stream())


def build():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _issue(ops, x, streams, 6)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with GS.Labels() as lab, torch.cuda.graph(g, capture_error_mode="thread_local"):
        origin = torch.cuda.current_stream().cuda_stream
        out = _issue(ops, x, streams, 6)
    return GS.SplitGraph(g, lab, origin), out


def run(tag, main_stream):
    with torch.cuda.stream(main_stream) if main_stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        sp, out = build()
        want = _issue(ops, x, streams, 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        e = sp.errors()
        print(f"{tag}: main stream {torch.cuda.current_stream().cuda_stream:#x} sides {[hex(s.cuda_stream) for s in sp._side]} "
              f"launch+sync {dt:.1f} ms, timed-out waits {e}, equal {torch.equal(out, want)}, info {sp.info}", flush=True)
        # epochs / flags after the run
        print("   words:", sp.words.tolist()[:12], flush=True)
        for c, k, near, fl, b, e in sp.timeline():
            print(f"     chain {c} {k:6s} near node {near:3d} flags {fl}  begin {b:9.1f} us  blocked {e - b:9.1f} us", flush=True)
        t0 = time.perf_counter()
        for _ in range(20):
            sp.launch()
        torch.cuda.synchronize()
        print(f"   20 more launches: {(time.perf_counter() - t0) * 50:.3f} ms each, errors {sp.errors()}, equal {torch.equal(out, want)}", flush=True)


def run2(tag, order):
    """explicit launches with host timing per call; order: list of chain ids"""
    ms = torch.cuda.Stream()
    with torch.cuda.stream(ms):
        sp, out = build()
        want = _issue(ops, x, streams, 6)
        torch.cuda.synchronize()
        sp.launch(); torch.cuda.synchronize()           # sets up exec streams
        from bist_amd import graphsplit
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        # reach into the object: launch chain by chain through the runtime to time each call
        arr = sp._arr
        for rep in range(2):
            torch.cuda.synchronize()
            e0 = sp.errors()
            ts = []
            t00 = time.perf_counter()
            sp.launch_order(order, ts)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{tag} order {order}: per-call host ms {[round(t * 1e3, 3) for t in ts]}, all calls {(t1 - t00) * 1e3:.3f} ms, done after {(t2 - t00) * 1e3:.3f} ms, new errors {sp.errors() - e0}, equal {torch.equal(out, want)}", flush=True)
        for c, k, near, fl, b, e in sp.timeline()[:14]:
            print(f"     chain {c} {k:6s} near node {near:3d} flags {fl}  begin {b:9.1f} us  blocked {e - b:9.1f} us", flush=True)


if os.environ.get("VARIANT", "0") == "0":
    run("null main", None)
    run("own main", torch.cuda.Stream())
else:
    run2("sides first", [1, 2, 0])
    run2("main first", [0, 1, 2])
