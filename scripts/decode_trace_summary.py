"""Spans of one beam-search turn out of a rocprofv3 kernel trace (results.db): first step, each later step, gaps (development aid).
usage: python scripts/decode_trace_summary.py <results.db>"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
short = lambda n: re.sub(r"^_ZN\d*_?GLOBAL__N_1\d+", "", n)[:44]
dec = [i for i, r in enumerate(rows) if "decstack" in r[0]]
sel = [i for i, r in enumerate(rows) if "beam_select" in r[0]]
# the last turn: its 12 decstack launches
d12 = dec[-12:]
prev_end = max(i for i in sel if i < d12[0] and i < dec[-12]) if len(dec) > 12 else 0
# the turn starts after the previous turn's last beam_select
start_i = [i for i in sel if i < d12[0]][-1] + 1 if [i for i in sel if i < d12[0]] else 0
t0 = rows[start_i][1]
print(f"turn: {len(rows[start_i:sel[-1] + 1])} launches, span {(rows[sel[-1]][2] - t0) / 1e3:.0f} us")
first_end = [i for i in sel if i > d12[0]][0]
print(f"first step: {first_end - start_i + 1} launches, span {(rows[first_end][2] - t0) / 1e3:.0f} us, of which before the decoder kernel {(rows[d12[0]][1] - t0) / 1e3:.0f} us")
busy = sum(r[2] - r[1] for r in rows[start_i:d12[0]]) / 1e3
print(f"  kernel time before the decoder kernel {busy:.0f} us (the rest is gaps / overlap)")
c = collections.Counter(); n = collections.Counter()
for r in rows[start_i:d12[0]]:
    c[short(r[0])] += (r[2] - r[1]) / 1e3; n[short(r[0])] += 1
for k, v in c.most_common(12): print(f"    {n[k]:4d} x {v / n[k]:6.1f} us = {v:7.0f}  {k}")
ends = [i for i in sel if i > d12[0]]
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    print(f"step: {len(seg)} launches, span {(seg[-1][2] - rows[a][2]) / 1e3:6.0f} us: " + ", ".join(f"{short(r[0])[:18]} {(r[2] - r[1]) / 1e3:.0f}" for r in seg))
    if a == ends[2]: break
