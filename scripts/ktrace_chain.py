"""One steady-state step of each hardware queue from a rocprofv3 kernel trace of scripts/bench_step.py under the split executor
(development aid): per queue the launches between its third-last and second-last epoch bump, with start, duration and the gap to the
previous launch of that queue.  usage: ktrace_chain.py <trace dir> [max launches of the queue to print]"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
steps = {}
for q, rs in byq.items():
    b = [k for k, r in enumerate(rs) if 'gs_bump' in r['Kernel_Name']]
    if len(b) >= 4:
        steps[q] = rs[b[-3]:b[-2]]
t0 = min(int(s[0]['Start_Timestamp']) for s in steps.values())
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').replace('_ZN12_GLOBAL__N_1', '').split('(')[0][:44]
for q, rs in sorted(steps.items(), key=lambda kv: len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs if 'gs_wait' not in r['Kernel_Name'])
    waits = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs if 'gs_wait' in r['Kernel_Name'])
    span = int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])
    print("queue %s: %d launches, span %.0f us, kernels %.0f us, inside waits %.0f us, neither %.0f us" % (q, len(rs), span / 1e3, busy / 1e3, waits / 1e3, (span - busy - waits) / 1e3))
q = min(steps, key=lambda q_: len(steps[q_]))
print("--- the shortest queue (%s), first %d launches" % (q, want))
prev = None
for r in steps[q][:want]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f %8.1f gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, short(r['Kernel_Name'])))
    prev = e
