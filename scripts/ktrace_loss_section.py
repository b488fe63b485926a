"""The serial sections of a training step from a rocprofv3 kernel trace taken with BIST_SPLIT_GRAPH=0 (the runtime's executor: durations are
not inflated by spinning waits): every launch between the last decoder-layer kernel of the forward pass and the first backward launch of the
layer stacks (the losses), and the launches after the last layer's backward (the tail).  usage: ktrace_loss_section.py <trace dir>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').replace('_ZN12_GLOBAL__N_1', '').split('(')[0][:60]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id']) for r in rows]
# one step: between the last two noam_hyper launches
heads = [i for i, e in enumerate(ev) if 'noam_hyper' in e[2]]
step = ev[heads[-2]:heads[-1]]
t0 = step[0][0]
ls = [i for i, e in enumerate(step) if 'label_smoothing' in e[2] or 'xent_smooth' in e[2]]
lo, hi = max(0, ls[0] - 25), min(len(step), ls[-1] + 40)
print("step: %d launches, %.1f us" % (len(step), (step[-1][1] - t0) / 1e3))
print("--- around the losses")
for s, e, n, q in step[lo:hi]:
    print("%9.1f %7.1f q%-2s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
print("--- the last 45 launches of the step")
for s, e, n, q in step[-45:]:
    print("%9.1f %7.1f q%-2s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
