# as prof_step_r03.sh at T = 128 (BASELINE configs[3]) -> gpurun_out/r03_step_t128_{one_step,timeline}.txt (development aid)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_t128
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t128 -- python3 bench.py --T 128 --steps 6 --warmup 3 --no-t128 --no-f32 --no-cpu-baseline --no-decode > gpurun_out/prof_t128.json 2> gpurun_out/prof_t128.err
T=$(find gpurun_out/prof_t128 -name "*kernel_trace.csv" | head -1)
python scripts/step_launch_counts.py $T > gpurun_out/r03_step_t128_one_step.txt 2>&1
python scripts/step_timeline.py $T 48 > gpurun_out/r03_step_t128_timeline.txt 2>&1
rm -rf gpurun_out/prof_t128
