"""Print the tail of a rocprofv3 kernel trace as (begin, end, duration, queue, kernel) relative to the Nth-last gemm_big launch (development aid)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'gemm_big' in r['Kernel_Name']]
start = idx[-back]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    b, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%9.1f %9.1f %7.1f q%-3s %s" % (b / 1e3, e / 1e3, (e - b) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
