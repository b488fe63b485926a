import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
gmin = min(int(rows[i]["Grid_Size_X"]) for i in ad)
ends = [i for i in ad if int(rows[i]["Grid_Size_X"]) == gmin] if len({int(rows[i]["Grid_Size_X"]) for i in ad}) > 1 else ad
seg = rows[ends[-2] + 1:ends[-1] + 1]
c = collections.Counter(); t = collections.Counter()
for r in seg:
    m = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"])
    name = (m.group(1) if m else r["Kernel_Name"][:50]).replace("_ZN12_GLOBAL__N_1", "")
    name = re.sub(r"^\d+", "", name)[:40]
    c[name] += 1; t[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(len(seg), "launches", sum(t.values()) / 1e3, "ms")
for k, v in c.most_common(45):
    print(f"{v:5d}  {t[k]:9.1f} us  {t[k]/v:7.1f} us each  {k}")
