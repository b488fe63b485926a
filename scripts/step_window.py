import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
gmin = min(int(rows[i]["Grid_Size_X"]) for i in ad)
ends = [i for i in ad if int(rows[i]["Grid_Size_X"]) == gmin] if len({int(rows[i]["Grid_Size_X"]) for i in ad}) > 1 else ad
seg = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
a, b = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
prev_end = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s < a or s > b: continue
    m = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"])
    name = (m.group(1) if m else r["Kernel_Name"][:40]).replace("_ZN12_GLOBAL__N_1", "")[:36]
    print(f"{s/1e3:9.1f} us  q{r['Queue_Id']}  {(e-s)/1e3:7.1f} us  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:>4} wg {r['Workgroup_Size_X']:>4}  {name}")
