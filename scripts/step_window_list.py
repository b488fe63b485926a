"""The launches of ONE replayed training step inside a time window, in start order, from a rocprofv3 --kernel-trace CSV (development
aid): start, queue, duration, gap since the previous launch ended ON ANY queue, grid.  usage: python scripts/step_window_list.py <csv> <ms0> <ms1>"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
m0, m1 = float(sys.argv[2]), float(sys.argv[3])
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
gmin = min(int(rows[i]["Grid_Size_X"]) for i in ad)
ends = [i for i in ad if int(rows[i]["Grid_Size_X"]) == gmin] if len({int(rows[i]["Grid_Size_X"]) for i in ad}) > 1 else ad
seg = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
last_end = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if m0 * 1e6 <= s <= m1 * 1e6:
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = re.sub(r"^\d+", "", (m.group(1) if m else r["Kernel_Name"][:30]).replace("_ZN12_GLOBAL__N_1", ""))[:26]
        wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        print(f"{s / 1e3:9.1f} us  q{r['Queue_Id']}  {(e - s) / 1e3:7.1f} us  idle-before {(s - last_end + t0) / 1e3 if False else (s - (last_end - t0)) / 1e3:7.1f}  wgs {wg:6d}  {name}")
    last_end = max(last_end, e + t0)
