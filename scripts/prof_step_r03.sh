set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for Z in 1 0; do
  export BIST_ZBATCH=$Z
  rm -rf gpurun_out/prof_z$Z
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_z$Z -- python3 bench.py --steps 8 --warmup 3 --no-t128 --no-cpu-baseline --no-decode > gpurun_out/prof_z$Z.json 2> gpurun_out/prof_z$Z.err
  T=$(find gpurun_out/prof_z$Z -name "*kernel_trace.csv" | head -1)
  python scripts/step_launch_counts.py $T > gpurun_out/r03_step_z${Z}_one_step.txt 2>&1
  python scripts/step_timeline.py $T 48 > gpurun_out/r03_step_z${Z}_timeline.txt 2>&1
  python scripts/queue_balance.py $T 3 > gpurun_out/r03_step_z${Z}_queues.txt 2>&1
  S=$(find gpurun_out/prof_z$Z -name "*kernel_stats.csv" | head -1)
  cp $S gpurun_out/r03_step_z${Z}_kernel_stats.csv
  rm -rf gpurun_out/prof_z$Z
  python -c "
import json
d=json.loads(open('gpurun_out/prof_z$Z.json').read().strip().splitlines()[-1]); print('Z=$Z step', d['ms_per_step'])"
done
unset BIST_ZBATCH
for Z in 1 0; do
  BIST_ZBATCH=$Z timeout -k 10 300 python -X faulthandler -m pytest tests/test_model_gpu.py -x -q -k "dialogue_lengths" > gpurun_out/r03_seg_z$Z.log 2>&1; echo "Z=$Z rc=$?"; tail -3 gpurun_out/r03_seg_z$Z.log
done
