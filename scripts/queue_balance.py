"""Busy time per hardware queue / stream over ONE replayed training step of a rocprofv3 kernel trace (development aid)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# the last occurrence of adam_kernel marks step ends; take the window between the last two adam launches of the replayed steps
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]          # adam_kernel / adam4_kernel / adam_apply4: every optimiser launch
if not adam:
    sys.exit("no optimiser launch in the trace: cannot cut it into steps")
if len({int(rows[i]["Grid_Size_X"]) for i in adam}) > 1:          # two Adam launches per step: the small (prefix) one ends it
    gmin = min(int(rows[i]["Grid_Size_X"]) for i in adam)
    adam = [i for i in adam if int(rows[i]["Grid_Size_X"]) == gmin]
if len(adam) < 2:
    sys.exit("fewer than two steps in the trace")
k = min(int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 1, len(adam) - 1)          # the k-th step (default: the last whole one)
lo, hi = adam[k - 1] + 1, adam[k] + 1
win = rows[lo:hi]
t0 = min(int(r["Start_Timestamp"]) for r in win); t1 = max(int(r["End_Timestamp"]) for r in win)
print(f"step window: {len(win)} launches, span {(t1 - t0) / 1e6:.2f} ms")
for key in ("Queue_Id", "Stream_Id"):
    busy = collections.Counter(); n = collections.Counter()
    for r in win:
        busy[r[key]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n[r[key]] += 1
    print(key, {q: (n[q], round(b, 2)) for q, b in sorted(busy.items())})
# concurrency profile: fraction of the span with 1, 2, 3+ kernels running
ev = []
for r in win:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(cur, 4)] += t - last; last = t; cur += d
tot = sum(hist.values())
print("time with k kernels in flight:", {k: f"{v / tot * 100:.1f}%" for k, v in sorted(hist.items())})
