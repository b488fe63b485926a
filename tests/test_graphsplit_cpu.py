"""The planner of the split-graph executor (csrc/graphsplit.hip: bist_graph_split_plan) on random DAGs, no device needed.

A "capture" is simulated: C streams issue nodes in a random interleaving; each node depends on its stream's previous node and, at random,
on the current tails of other streams (an event wait).  The plan must keep every chain in capture order and imply EVERY edge of the DAG
through chain order and signal -> wait pairs (happens-before closure of the plan), drop redundant waits, and never wait for a flag
that nothing signals."""
import ctypes as C
import random

import pytest

from bist_amd._lib import lib


def _capture(rng, n, streams, p_cross, p_unlabelled=0.0):
    tails = [None] * streams
    edges, labels, true = [], [], []
    for v in range(n):
        s = rng.randrange(streams) if v else 0
        if tails[s] is None and v:                        # a fork: the stream starts behind some other stream's tail
            src = rng.choice([t for t in tails if t is not None])
            edges.append((src, v))
        elif tails[s] is not None:
            edges.append((tails[s], v))
        for o in range(streams):
            if o != s and tails[o] is not None and rng.random() < p_cross and (tails[o], v) not in edges:
                edges.append((tails[o], v))
        tails[s] = v
        true.append(s)
        labels.append(-1 if rng.random() < p_unlabelled else s)
    return edges, labels, true


def _plan(n, edges, labels, chains, main=0):
    ef = (C.c_int32 * max(1, len(edges)))(*[e[0] for e in edges])
    et = (C.c_int32 * max(1, len(edges)))(*[e[1] for e in edges])
    lb = (C.c_int32 * n)(*labels)
    need = lib.bist_graph_split_plan(n, ef, et, len(edges), lb, chains, main, None, 0)
    assert need > 0, lib.bist_last_error().decode()
    out = (C.c_int32 * need)()
    assert lib.bist_graph_split_plan(n, ef, et, len(edges), lb, chains, main, out, need) == need
    flat, seqs, i = list(out), {}, 0
    while flat[i] != -2:
        assert flat[i] == -1
        c = flat[i + 1]; i += 2
        seqs[c] = []
        while flat[i] not in (-1, -2):
            seqs[c].append(tuple(flat[i:i + 6])); i += 6
    return seqs, flat[i + 1]


def _check(n, edges, seqs, n_flags, main=0):
    # every node exactly once; chains in index (capture) order
    where = {}
    for c, s in seqs.items():
        last = -1
        assert s[0][0] == 1, "a chain starts with its epoch bump"
        for k, it in enumerate(s):
            if it[0] == 0:
                assert it[1] not in where and it[1] > last
                where[it[1]] = (c, k); last = it[1]
    assert len(where) == n
    # happens-before: per item the set of nodes known to have finished, as a bitmask; signals publish theirs on the flag
    signal_of = {}
    for c, s in seqs.items():
        for k, it in enumerate(s):
            if it[0] == 2:
                assert it[1] not in signal_of, "one signaller per flag"
                signal_of[it[1]] = (c, k)
    known_at_flag = {}
    progress, state = True, {c: [None] * len(s) for c, s in seqs.items()}
    done = {c: 0 for c in seqs}
    acc = {c: 0 for c in seqs}
    while progress:                                   # run the chains like the device would: a wait blocks until its flags are signalled
        progress = False
        for c, s in seqs.items():
            while done[c] < len(s):
                it = s[done[c]]
                if it[0] == 3:
                    fl = [f for f in it[2:6] if f >= 0]
                    assert fl and all(f in signal_of and f < n_flags for f in fl), "a wait for a flag nobody signals"
                    if not all(f in known_at_flag for f in fl):
                        break
                    for f in fl:
                        acc[c] |= known_at_flag[f]
                elif it[0] == 2:
                    known_at_flag[it[1]] = acc[c]
                elif it[0] == 0:
                    state[c][done[c]] = acc[c]        # what is known to have finished when this node starts
                    acc[c] |= 1 << it[1]
                done[c] += 1; progress = True
    assert all(done[c] == len(s) for c, s in seqs.items()), "the plan deadlocks"
    for u, v in edges:
        c, k = where[v]
        assert state[c][k] >> u & 1, f"edge {u}->{v} is not implied by the plan"
    # side chains start behind START (flag 0) and end with a signal; the main chain ends waiting for them
    for c, s in seqs.items():
        if c == main:
            assert s[1][:2] == (2, 0)
            if len(seqs) > 1:
                assert s[-1][0] == 3
        else:
            assert s[1][0] == 3 and s[1][2] == 0 and s[-1][0] == 2


@pytest.mark.parametrize("seed", range(12))
def test_random_captures_are_fully_ordered_by_the_plan(seed):
    rng = random.Random(seed)
    streams = rng.choice([1, 2, 3, 3, 4, 6])
    n = rng.choice([1, 5, 40, 300])
    edges, labels, _ = _capture(rng, n, streams, p_cross=rng.choice([0.0, 0.1, 0.5]))
    seqs, n_flags = _plan(n, edges, labels, streams)
    _check(n, edges, seqs, n_flags)


def test_unlabelled_nodes_join_a_predecessors_chain_and_waits_are_pruned():
    rng = random.Random(7)
    edges, labels, true = _capture(rng, 400, 3, p_cross=0.3, p_unlabelled=0.15)
    seqs, n_flags = _plan(400, edges, labels, 3)
    _check(400, edges, seqs, n_flags)
    waits = sum(1 for s in seqs.values() for it in s if it[0] == 3)
    cross = sum(1 for u, v in edges if true[u] != true[v])
    assert waits < cross, (waits, cross)              # vector clocks drop the implied ones


def test_a_chain_of_one_stream_needs_no_sync_but_its_epoch():
    seqs, n_flags = _plan(6, [(i, i + 1) for i in range(5)], [0] * 6, 1)
    assert [it[0] for it in seqs[0]] == [1, 2, 0, 0, 0, 0, 0, 0] and n_flags == 1


def test_bad_input_is_refused():
    ef, et, lb = (C.c_int32 * 1)(0), (C.c_int32 * 1)(1), (C.c_int32 * 2)(0, 5)
    assert lib.bist_graph_split_plan(2, ef, et, 1, lb, 2, 0, None, 0) == -1           # label >= chains
    lb2 = (C.c_int32 * 2)(0, 0)
    ef2, et2 = (C.c_int32 * 2)(0, 1), (C.c_int32 * 2)(1, 0)
    assert lib.bist_graph_split_plan(2, ef2, et2, 2, lb2, 1, 0, None, 0) == -1        # cycle
    assert lib.bist_graph_split_plan(2, ef, et, 1, lb2, 9, 0, None, 0) == -1          # too many chains
