"""Coarse timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV (development aid): per time bucket the busy
share of each hardware queue and the kernels that dominate it.  usage: python scripts/step_timeline.py <kernel_trace.csv> [buckets]"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
gmin = min(int(rows[i]["Grid_Size_X"]) for i in ad)
ends = [i for i in ad if int(rows[i]["Grid_Size_X"]) == gmin] if len({int(rows[i]["Grid_Size_X"]) for i in ad}) > 1 else ad
seg = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
w = (t1 - t0) / nb
queues = sorted({r["Queue_Id"] for r in seg})
busy = [collections.Counter() for _ in range(nb)]
top = [collections.Counter() for _ in range(nb)]
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    name = re.sub(r"^\d+", "", (m.group(1) if m else r["Kernel_Name"][:24]).replace("_ZN12_GLOBAL__N_1", ""))[:22]
    b0, b1 = int(s // w), min(nb - 1, int(e // w))
    for b in range(b0, b1 + 1):
        ov = min(e, (b + 1) * w) - max(s, b * w)
        if ov > 0:
            busy[b][r["Queue_Id"]] += ov
            top[b][name] += ov
print(f"span {(t1 - t0) / 1e6:.2f} ms, {len(seg)} launches, bucket {w / 1e3:.0f} us; columns: busy share per queue {queues}")
for b in range(nb):
    print(f"{b * w / 1e6:6.2f} ms  " + " ".join(f"{busy[b][q] / w:4.2f}" for q in queues) + "   " + ", ".join(f"{k} {v / w:.2f}" for k, v in top[b].most_common(3)))
