# rocprofv3 kernel trace of the bench's training step -> one-step launch counts, timeline, queue balance (development aid)
# usage: bash scripts/prof_step_r03.sh <tag> [ENV=VALUE ...]
set -x
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 8 --warmup 3 --no-t128 --no-f32 --no-cpu-baseline --no-decode > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python scripts/step_launch_counts.py $T > gpurun_out/r03_step_${TAG}_one_step.txt 2>&1
python scripts/step_timeline.py $T 48 > gpurun_out/r03_step_${TAG}_timeline.txt 2>&1
python scripts/queue_balance.py $T 3 > gpurun_out/r03_step_${TAG}_queues.txt 2>&1
S=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/r03_step_${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
python -c "
import json
d=json.loads(open('gpurun_out/prof_$TAG.json').read().strip().splitlines()[-1]); print('$TAG step (profiled)', d['ms_per_step'])"
