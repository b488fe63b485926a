"""Per hardware queue, the ordered launches of ONE steady-state training step from a rocprofv3 kernel trace of scripts/bench_step.py with the
split-graph executor (one queue per chain): kernel time, gap to the previous launch of the queue.  usage: chain_digest.py trace.csv [layer_marker]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bump = [i for i, r in enumerate(rows) if "gs_bump" in r["Kernel_Name"]]
# a step = 3 bumps; take the window from the 3rd-last group's first bump to the last group's first bump
groups = [bump[i:i + 3] for i in range(0, len(bump) - len(bump) % 3, 3)]
g0, g1 = groups[-3][0], groups[-2][0]
t_lo, t_hi = int(rows[g0]["Start_Timestamp"]), int(rows[g1]["Start_Timestamp"])
win = [r for r in rows if t_lo <= int(r["Start_Timestamp"]) < t_hi]
print(f"step window: {len(win)} launches, {(t_hi - t_lo) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    sync = sum(1 for r in rs if "gs_" in r["Kernel_Name"])
    waitt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if "gs_wait" in r["Kernel_Name"]) / 1e3
    print(f"queue {q}: {len(rs)} launches ({sync} sync), kernel time {busy:.0f} us of which waits {waitt:.0f}, span {span:.0f} us, idle between launches {span - busy:.0f} us")
which = sys.argv[2] if len(sys.argv) > 2 else None
if which:
    rs = byq[which]
    prev = None
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = e
        print(f"{(s - t_lo) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f}  {short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
