"""Per-kernel means of rocprofv3 --pmc counters (development aid): python scripts/pmc_by_kernel.py <counter_collection.csv> [name filter]"""
import collections
import csv
import re
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    m = re.search(r"(\w+_kernel)", n)
    key = (m.group(1) if m else n[:40], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key, {k: round(sum(v) / len(v), 1) for k, v in cs.items()}, "launches", len(next(iter(cs.values()))))
