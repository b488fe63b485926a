# Per-kernel statistics of the training step WITHOUT the split executor's wait launches in the way of the profiler (the kernels are the same;
# under rocprofv3 the waits spin while the profiler serialises, which inflates every duration): BIST_SPLIT_GRAPH=0, runtime executor.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pr_step
BIST_SPLIT_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr_step -- python3 bench.py --steps 8 --warmup 3 --no-t128 --no-f32 --no-cpu-baseline --no-decode --no-fed > $O/pr_step.json 2> $O/pr_step.err || exit 1
T=$(find $O/pr_step -name "*kernel_trace.csv" | head -1)
python scripts/step_launch_counts.py $T > $O/r04_train_step_one_step.txt 2>&1
python scripts/queue_balance.py $T > $O/r04_train_step_queues.txt 2>&1
cp $(find $O/pr_step -name "*kernel_stats.csv" | head -1) $O/r04_train_step_kernel_stats.csv
rm -rf $O/pr_step
head -14 $O/r04_train_step_one_step.txt; cat $O/r04_train_step_queues.txt
