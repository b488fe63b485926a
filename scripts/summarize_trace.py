"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace CSV (development aid; its output is what
profiles/*_by_grid.txt hold).  usage: python scripts/summarize_trace.py <kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)", n)
    short = m.group(1) if m else n[:40]
    t = re.search(r"Lb(\d)ELb(\d)ELi(\d)", n)
    if t:
        short += "<A%s,B%s,%sst>" % ("T" if t.group(1) == "1" else "N", "T" if t.group(2) == "1" else "N", t.group(3))
    wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
    agg[(short, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} launches ({tot/1e3/steps:.2f} ms/step at {steps:g} steps)")
print(f"{'kernel':46s} {'WGs':>7s} {'n/step':>7s} {'avg us':>8s} {'min us':>8s} {'max us':>8s} {'ms/step':>8s}")
for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print(f"{k:46s} {w:7d} {len(v)/steps:7.1f} {sum(v)/len(v):8.1f} {min(v):8.1f} {max(v):8.1f} {sum(v)/1e3/steps:8.3f}")
