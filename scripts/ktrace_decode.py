"""Timeline of the LAST beam-search turn in a rocprofv3 kernel trace of scripts/bench_decode_r04.py (development aid): where the turn's
6.5 ms go -- the first step (encoders, input projection, six reasoning layers at B = 1, the decoder's first position) and the later steps
(persistent decoder kernel, generator, beam kernels) with the idle time between consecutive kernels of the whole device."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows
      if 'gs_idle' not in r['Kernel_Name']]
starts = [i for i, e in enumerate(ev) if 'stage_inputs' in e[2]]
i0 = starts[-1]
turn = ev[i0:]
t0 = turn[0][0]
dec = [e for e in turn if 'decstack' in e[2]]
print("turn: %d kernels, %.1f us from the input staging to the last kernel's end" % (len(turn), (max(e[1] for e in turn) - t0) / 1e3))
first_dec = dec[0][0]
print("first step (until the first persistent-decoder launch): %.1f us, %d kernels" % ((first_dec - t0) / 1e3, sum(1 for e in turn if e[0] < first_dec)))
# steps: boundaries at decstack<false/true> pairs
bounds = sorted({e[0] for e in dec})
print("persistent decoder launches: %d; durations us: %s" % (len(dec), [round((e[1] - e[0]) / 1e3) for e in dec]))
# union busy time
def busy(evs):
    iv = sorted((e[0], e[1]) for e in evs if 'gs_wait' not in e[2])
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
pre = [e for e in turn if e[0] < first_dec]
post = [e for e in turn if e[0] >= first_dec]
print("first step: device busy (any kernel but waits) %.1f us of %.1f" % (busy(pre) / 1e3, (first_dec - t0) / 1e3))
print("later steps: busy %.1f us of %.1f" % (busy(post) / 1e3, (max(e[1] for e in post) - first_dec) / 1e3))
from collections import defaultdict
for name, part in (("first step", pre), ("later steps", post)):
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, q in part:
        k = n.split('(')[0][-60:]
        agg[k][0] += 1; agg[k][1] += e - s
    print("---", name)
    for k, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %4d x %8.1f us = %9.1f us  %s" % (cnt, ns / cnt / 1e3, ns / 1e3, k))
if len(sys.argv) > 2:
    for s, e, n, q in pre:
        print("%9.1f %9.1f %7.1f q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:80]))
