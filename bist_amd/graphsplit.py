"""Host side of the split-graph executor (csrc/graphsplit.hip): a step captured over several HIP streams is replayed as one LINEAR
hipGraph per stream, the streams tied together by device-side flag words instead of by the runtime's multi-branch graph executor.

    g = torch.cuda.CUDAGraph(keep_graph=True)
    with Labels() as lab, Fn.capture_graph(g):
        ... the step, issued over torch streams ...
    split = SplitGraph(g, lab)          # plans, clones, prunes, instantiates (the torch graph object keeps the capture's memory alive)
    split.launch()                      # per step: one hipGraphLaunch per chain

``Labels`` hooks ``_lib.check`` -- which follows every launch of this library -- and asks the runtime for the capturing stream's newest
node: "node X was captured on stream S".  Launches the hook does not see (a few framework-internal fills and copies per step) are
placed by the planner on the chain of a predecessor.

Every chain is launched INTO THE STREAM IT WAS CAPTURED ON.  Measured on this runtime (profiles/r04_split_queue_aliasing.txt): the
kernel nodes of a clone of a stream-captured graph run on the stream they were captured on, whatever stream the clone is launched
into -- only the sync launches this module adds follow the launch stream.  Launched elsewhere, a chain's captured nodes land behind
another chain's wait on that queue, which can then only time out.  So the capture streams themselves must sit on pairwise different
hardware queues: ``distinct_streams(n)`` hands out such a set (probed with ``bist_graph_queues_distinct``), and callers capture on
it -- the capturing stream included (``torch.cuda.graph(..., stream=...)``), since a chain must never run on the NULL stream, which
waits for every other stream of the device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import check, lib

TIMEOUT_TICKS = int(float(os.environ.get("BIST_SPLIT_TIMEOUT_MS", "2000")) * 1e5)      # per wait launch, 100 MHz device clock


class Labels:
    """``with Labels() as lab:`` around a capture: lab.node_stream[node handle] = handle of the stream the node was captured on."""

    def __init__(self):
        self.node_stream: Dict[int, int] = {}
        self.origin: Optional[int] = None
        self._prev = None

    def _hook(self) -> None:
        st = torch.cuda.current_stream().cuda_stream
        node = C.c_void_p()
        if lib.bist_graph_capture_tail(st, C.byref(node)) == 0 and node.value:
            self.node_stream[node.value] = st

    def note_stream(self, stream) -> None:
        """Remember that `stream` takes part (a stream none of whose nodes the hook saw still gets no chain)."""
        self._hook()

    def __enter__(self):
        self._prev = _lib.AFTER_LAUNCH
        _lib.AFTER_LAUNCH = self._hook
        return self

    def __exit__(self, *exc):
        _lib.AFTER_LAUNCH = self._prev
        return False


_EXEC_STREAMS: Dict[int, List[torch.cuda.Stream]] = {}


def _distinct(a: int, b: int, scratch: torch.Tensor) -> bool:
    scratch.zero_()
    torch.cuda.synchronize()
    check(lib.bist_graph_queues_distinct(a, b, scratch.data_ptr(), int(5e5)), "bist_graph_queues_distinct")      # 5 ms
    torch.cuda.synchronize()
    return int(scratch[1].item()) == 0


def _pace(a: int, b: Optional[int], scratch: torch.Tensor) -> float:
    """us per launch of a 200-launch single-queue graph on stream a, alone or beside a wave resident on stream b."""
    out = C.c_float()
    torch.cuda.synchronize()
    check(lib.bist_graph_queue_pace(a, 200, b, int(2e-3 * 1e8), scratch.data_ptr(), C.byref(out)), "bist_graph_queue_pace")
    return float(out.value)


def _independent(a: int, b: int, scratch: torch.Tensor) -> bool:
    """Different hardware queues AND different dispatch pipes: a wave resident on one does not slow the other's launches."""
    if not (_distinct(a, b, scratch) and _distinct(b, a, scratch)):
        return False
    alone = min(_pace(a, None, scratch), _pace(a, None, scratch))
    return _pace(a, b, scratch) < 1.6 * alone and _pace(b, a, scratch) < 1.6 * min(_pace(b, None, scratch), _pace(b, None, scratch))


def distinct_streams(n: int) -> List[torch.cuda.Stream]:
    """n streams (none of them the NULL stream) on pairwise different hardware queues that do not share a dispatch pipe: probed (a wait
    ahead of its signal must see it; a resident wave on one must not slow the launches of the other), kept per device, the same set
    for every caller.  MI355X has four such pipes.  Must not be called during a capture (the probe synchronises)."""
    dev = torch.cuda.current_device()
    keep = _EXEC_STREAMS.setdefault(dev, [])
    if len(keep) >= n:
        return keep[:n]
    scratch = torch.zeros(4, dtype=torch.int64, device="cuda")
    tries = 0
    while len(keep) < n:
        tries += 1
        if tries > 96:
            raise RuntimeError(f"bist_amd.graphsplit: no {n} streams on hardware queues of their own "
                               "(DEBUG_HIP_DYNAMIC_QUEUES=0 and GPU_MAX_HW_QUEUES >= 8 before the first HIP call make that certain)")
        s = torch.cuda.Stream()
        if any(s.cuda_stream == o.cuda_stream for o in keep):
            continue
        if all(_independent(o.cuda_stream, s.cuda_stream, scratch) for o in keep):
            keep.append(s)
    return keep[:n]


def queue_distinct_stream(avoid: List[torch.cuda.Stream], pipe_with: Optional[torch.cuda.Stream] = None) -> torch.cuda.Stream:
    """A stream on a hardware QUEUE none of `avoid` uses (it shares a dispatch pipe with one of them: all four carry a chain): for work that
    must never sit in a chain's queue -- a host-to-device copy and what is ordered behind it hold their queue for milliseconds.
    pipe_with: the stream whose pipe the new stream should share -- while the copy is in flight that stream's launches pay the threefold
    gap, so the caller names the chain with slack (a training step's caption chain: 170 of 1 165 launches); first choice, not a condition.
    Not during a capture (the probe synchronises)."""
    scratch = torch.zeros(4, dtype=torch.int64, device="cuda")
    alone = {o.cuda_stream: min(_pace(o.cuda_stream, None, scratch) for _ in range(3)) for o in avoid} if pipe_with is not None else {}
    fallback = None
    for _ in range(96):
        s = torch.cuda.Stream()
        if any(s.cuda_stream == o.cuda_stream for o in avoid) or (fallback is not None and s.cuda_stream == fallback.cuda_stream):
            continue
        if not all(_distinct(o.cuda_stream, s.cuda_stream, scratch) and _distinct(s.cuda_stream, o.cuda_stream, scratch) for o in avoid):
            continue
        if pipe_with is None:
            return s
        # whose launches does a wave resident on s slow down?  (each of the remaining queues is the pipe partner of exactly one chain)
        ratio = {o.cuda_stream: min(_pace(o.cuda_stream, s.cuda_stream, scratch) for _ in range(3)) / alone[o.cuda_stream] for o in avoid}
        slowed = [k for k, r in ratio.items() if r >= 1.5]
        if os.environ.get("BIST_GS_DEBUG"):
            print("graphsplit.queue_distinct_stream: candidate %#x slows %s" % (s.cuda_stream, {hex(k): round(r, 2) for k, r in ratio.items()}), flush=True)
        if slowed == [pipe_with.cuda_stream]:
            return s
        fallback = fallback or s
    if fallback is not None:
        return fallback
    raise RuntimeError("bist_amd.graphsplit: no stream on a hardware queue of its own beside the chains' (GPU_MAX_HW_QUEUES >= 8 before the first HIP call)")


_USABLE: Dict[int, bool] = {}
WHY_NOT = ""            # why usable() said no (diagnostics)


def usable() -> bool:
    """May this process replay split graphs?  The executor needs the HIP runtime's static stream -> queue mapping and more than four
    hardware queues (DEBUG_HIP_DYNAMIC_QUEUES=0, GPU_MAX_HW_QUEUES >= 6: set by `import bist_amd` when the runtime is not yet
    initialised).  Whether they were in place in time cannot be read back, so a three-stream graph with waits ahead of their signals
    is replayed once per device with a 5 ms time-out: with the dynamic mapping a chain's later packets land behind another chain's
    wait (the failure measured in profiles/r04_split_queue_aliasing.txt) and waits time out.  False -> callers replay through the
    runtime's own executor."""
    dev = torch.cuda.current_device()
    if dev in _USABLE:
        return _USABLE[dev]
    ok = (os.environ.get("DEBUG_HIP_DYNAMIC_QUEUES") == "0" and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 6
          and os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES") == "1")
    global WHY_NOT
    if not ok:
        WHY_NOT = "DEBUG_HIP_DYNAMIC_QUEUES=0, GPU_MAX_HW_QUEUES>=6 and DEBUG_HIP_FORCE_GRAPH_QUEUES=1 were not in the environment"
    else:
        try:
            ok = _self_test()
            if not ok:
                WHY_NOT = "the self-test's waits timed out (the runtime was initialised before the switches were set?)"
        except Exception as e:          # noqa: BLE001  (anything: the caller falls back to the runtime's executor)
            ok, WHY_NOT = False, "self-test raised %s: %s" % (type(e).__name__, e)
    _USABLE[dev] = ok
    return ok


def _self_test() -> bool:
    from . import ops
    x = torch.full((4096,), 0.5, device="cuda")
    s0, s1, s2 = distinct_streams(3)

    def issue():
        main = torch.cuda.current_stream()
        acc = ops.add(x, x)
        for _ in range(3):
            a = acc
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                b = ops.add(a, x)
                b = ops.add(b, x)
            with torch.cuda.stream(s2):
                c = ops.add(a, a)
            d = ops.add(a, x)
            main.wait_stream(s1); main.wait_stream(s2)
            acc = ops.add_n([b, c, d])
        return acc
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        want = issue()
    torch.cuda.current_stream().wait_stream(warm)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with Labels() as lab, torch.cuda.graph(g, stream=s0, capture_error_mode="thread_local"):
        origin = torch.cuda.current_stream().cuda_stream
        out = issue()
    sp = SplitGraph(g, lab, origin, timeout_ticks=int(2e6))          # 20 ms per wait
    sp.launch()                       # (the first launch of an exec may take the host milliseconds: its time-outs do not count)
    torch.cuda.synchronize()
    sp.words[sp.n_chains].zero_()
    for _ in range(3):
        sp.launch()
    ok = sp.errors() == 0 and bool(torch.equal(out, want))
    if not ok and os.environ.get("BIST_SPLIT_DEBUG"):
        print("graphsplit self-test: errors", sp.errors(), "equal", bool(torch.equal(out, want)), "info", sp.info,
              "streams", [hex(h_) for h_ in sp.capture_streams], flush=True)
        for c, k, near, fl, b, e in sp.timeline():
            print(f"     chain {c} {k:6s} near node {near:3d} flags {fl}  begin {b:9.1f} us  blocked {e - b:9.1f} us", flush=True)
    return ok


class SplitGraph:
    def __init__(self, graph: torch.cuda.CUDAGraph, labels: Labels, origin_stream: int, timeout_ticks: int = TIMEOUT_TICKS):
        self.graph = graph                                   # owns the capture's memory pool
        raw = graph.raw_cuda_graph()
        n = C.c_int32()
        check(lib.bist_graph_nodes(raw, None, 0, C.byref(n)), "bist_graph_nodes")
        nodes = (C.c_void_p * n.value)()
        check(lib.bist_graph_nodes(raw, nodes, n.value, C.byref(n)), "bist_graph_nodes")
        streams = sorted(set(labels.node_stream.values()) | {origin_stream}, key=lambda s: (s != origin_stream, s))
        chain_of = {s: i for i, s in enumerate(streams)}
        self.capture_streams = streams
        lab = (C.c_int32 * n.value)(*[chain_of.get(labels.node_stream.get(nodes[i], -1), -1) for i in range(n.value)])
        self.n_nodes, self.n_chains = n.value, len(streams)
        self.n_labelled = sum(1 for i in range(n.value) if lab[i] >= 0)
        h = C.c_void_p()
        check(lib.bist_graph_split_create(raw, lab, n.value, self.n_chains, 0, C.byref(h)), "bist_graph_split_create")
        self._h = h
        words = lib.bist_graph_split_sync_words(h)
        self.words = torch.zeros(words, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        check(lib.bist_graph_split_build(h, raw, self.words.data_ptr(), timeout_ticks), "bist_graph_split_build")
        info, per = (C.c_int32 * 5)(), (C.c_int32 * self.n_chains)()
        check(lib.bist_graph_split_info(h, info, per), "bist_graph_split_info")
        self.info = {"chains": info[0], "nodes": info[1], "waits": info[2], "signals": info[3], "flags": info[4], "nodes_per_chain": list(per),
                     "labelled": self.n_labelled}
        self._side: Optional[List[torch.cuda.Stream]] = None
        self._exec_main: Optional[torch.cuda.Stream] = None
        self._arr = (C.c_void_p * self.n_chains)()

    def launch(self) -> None:
        """One replay, ordered behind everything queued on the current stream; the current stream continues behind the replay.
        Every chain goes into the stream it was captured on (see the module docstring)."""
        cur = torch.cuda.current_stream()
        if self._exec_main is None:
            if self.capture_streams[0] == 0:
                raise RuntimeError("bist_amd.graphsplit: the graph was captured on the NULL stream (capture with torch.cuda.graph(..., stream=s))")
            ext = [torch.cuda.ExternalStream(h) for h in self.capture_streams]
            self._exec_main, self._side = ext[0], ext[1:]
            for i, s in enumerate(ext):
                self._arr[i] = s.cuda_stream
        same = cur.cuda_stream == self._exec_main.cuda_stream
        if not same:
            self._exec_main.wait_stream(cur)
        check(lib.bist_graph_split_launch(self._h, self._arr), "bist_graph_split_launch")
        if not same:
            cur.wait_stream(self._exec_main)

    def dump(self, path: str) -> None:
        """The plan (chain sequences) and the captured graph's edges as JSON (analysis aid: scripts/critical_path.py)."""
        import json
        n = lib.bist_graph_split_dump(self._h, None, 0)
        arr = (C.c_int32 * n)()
        lib.bist_graph_split_dump(self._h, arr, n)
        raw = self.graph.raw_cuda_graph()
        ne = C.c_int32()
        check(lib.bist_graph_edges(raw, None, None, 0, C.byref(ne)), "bist_graph_edges")
        ef, et = (C.c_int32 * max(1, ne.value))(), (C.c_int32 * max(1, ne.value))()
        check(lib.bist_graph_edges(raw, ef, et, ne.value, C.byref(ne)), "bist_graph_edges")
        with open(path, "w") as f:
            json.dump({"plan": list(arr), "edges": [[ef[i], et[i]] for i in range(ne.value)], "info": self.info}, f)

    def launch_order(self, order, times=None) -> None:
        """Development aid: the chains launched one by one in `order` (host seconds per call appended to `times`)."""
        import time
        for c in order:
            t0 = time.perf_counter()
            check(lib.bist_graph_split_launch_chain(self._h, c, self._arr[c]), "bist_graph_split_launch_chain")
            if times is not None:
                times.append(time.perf_counter() - t0)

    def timeline(self):
        """[(chain, kind, node index near it, flags, begin us, end us)] of the sync launches of the most recent replay, by begin time (us from
        the first): the device clock every epoch bump / signal / wait left in its stamp words.  A wait's end - begin is how long its chain
        stood blocked there.  Synchronises."""
        torch.cuda.synchronize()
        w = self.words.cpu().tolist()
        n = lib.bist_graph_split_sync_items(self._h, None, 0)
        arr = (C.c_int32 * n)()
        lib.bist_graph_split_sync_items(self._h, arr, n)
        base = self.n_chains + 1 + self.info["flags"]
        rows = []
        for i in range(n // 8):
            c, kind, near, f0, f1, f2, f3, _ = arr[8 * i:8 * i + 8]
            rows.append((c, {1: "bump", 2: "signal", 3: "wait"}[kind], near, [f for f in (f0, f1, f2, f3) if f >= 0], w[base + 2 * i], w[base + 2 * i + 1]))
        t0 = min(r[4] for r in rows if r[4] > 0)
        return sorted(((c, k, near, fl, (b - t0) / 100.0, (e - t0) / 100.0) for c, k, near, fl, b, e in rows), key=lambda r: r[4])

    def errors(self) -> int:
        """Waits that timed out so far (synchronises): anything but 0 voids every step since the last check."""
        torch.cuda.synchronize()
        return int(self.words[self.n_chains].item())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                torch.cuda.synchronize()
                lib.bist_graph_split_destroy(h)
            except Exception:
                pass
