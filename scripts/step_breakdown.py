"""Per-kernel breakdown of ONE training step (the launches between the last two Adam kernels) of a rocprofv3
--kernel-trace CSV (development aid).  usage: python scripts/step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
# a step ends with its LAST Adam launch: one launch per step, or two (big matrices beside the closing reductions, then the prefix)
gmin = min(int(rows[i]["Grid_Size_X"]) for i in ad)
ends = [i for i in ad if int(rows[i]["Grid_Size_X"]) == gmin] if len({int(rows[i]["Grid_Size_X"]) for i in ad}) > 1 else ad
seg = rows[ends[-2] + 1:ends[-1] + 1]
agg = collections.defaultdict(list)
for r in seg:
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)", n)
    short = re.sub(r"^\d+", "", (m.group(1) if m else n[:40]).replace("_ZN12_GLOBAL__N_1", ""))
    w = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(
        1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    agg[(short, w)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
print(f"one step: {len(seg)} launches, kernel time {tot/1e3:.2f} ms, span {span:.2f} ms")
byname = collections.defaultdict(lambda: [0, 0.0])
for (k, w), v in agg.items():
    byname[k][0] += len(v); byname[k][1] += sum(v)
print(f"{'kernel':40s} {'n':>5s} {'ms':>8s} {'avg us':>8s}")
for k, (n, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{k:40s} {n:5d} {t/1e3:8.3f} {t/n:8.1f}")
print()
print(f"{'kernel':40s} {'WGs':>7s} {'n':>5s} {'avg us':>8s} {'min us':>8s} {'ms':>8s}")
for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{k:40s} {w:7d} {len(v):5d} {sum(v)/len(v):8.1f} {min(v):8.1f} {sum(v)/1e3:8.3f}")
