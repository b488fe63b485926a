# as prof_step_r03.sh, but keeps the kernel trace CSV (gpurun_out/r03_step_<tag>_trace.csv) for window analyses (development aid)
set -x
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 6 --warmup 3 --no-t128 --no-f32 --no-cpu-baseline --no-decode > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python - "$T" gpurun_out/r03_step_${TAG}_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
rows = rows[ad[-8]:] if len(ad) > 8 else rows          # the last few steps are enough
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
rm -rf gpurun_out/prof_$TAG
