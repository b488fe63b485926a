"""Per-kernel durations of ONE pass over the roofline region from a rocprofv3 --kernel-trace CSV of scripts/prof_attn.py (the
dispatches between its two spin-kernel markers): python scripts/region_kernels.py <kernel_trace.csv> <iters>  ->  CSV on stdout"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
iters = int(sys.argv[2])
marks = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower()]
assert len(marks) >= 2
rows = rows[marks[-2] + 1:marks[-1]]
agg = collections.defaultdict(list)
for r in rows:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    agg[(m.group(1) if m else r["Kernel_Name"][:48], wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values()) / iters
print("kernel,workgroups,launches_per_pass,avg_us,min_us,max_us,us_per_pass,share")
for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k},{w},{len(v) / iters:.1f},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v) / iters:.2f},{sum(v) / iters / tot:.4f}")
print(f"TOTAL,,{sum(len(v) for v in agg.values()) / iters:.1f},,,,{tot:.2f},1.0")
