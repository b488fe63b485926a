# Round-4 evidence in one GPU call (run on the GPU box from the repo root): the roofline region's kernel trace + the two HBM counter
# passes, the training step's trace (launch counts, queues), the decode turn's trace.  Summaries land in gpurun_out/ (copy the ones to
# keep into profiles/).  usage: bash scripts/prof_round_r04.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
export BIST_SPLIT_TIMEOUT_MS=5000          # (under the profiler a chain may stand in a wait for long)
rm -rf $O/pr_region $O/pr_fetch $O/pr_write $O/pr_dec $O/pr_step
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr_region -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_region.log 2>&1 || exit 1
T=$(find $O/pr_region -name "*kernel_trace.csv" | head -1)
python scripts/region_kernels.py $T 5 > $O/r04_attn_fwd_B64_kernel_stats.csv
cp $(find $O/pr_region -name "*kernel_stats.csv" | head -1) $O/r04_attn_fwd_B64_rocprofv3_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pr_fetch -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pr_write -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_write.log 2>&1 || exit 1
python scripts/pmc_region.py $(find $O/pr_fetch -name "*counter_collection.csv" | head -1) $(find $O/pr_write -name "*counter_collection.csv" | head -1) 64 32 5 > $O/r04_attn_fwd_B64_pmc.json
rm -rf $O/pr_region $O/pr_fetch $O/pr_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr_step -- python3 bench.py --steps 8 --warmup 3 --no-t128 --no-f32 --no-cpu-baseline --no-decode --no-fed > $O/pr_step.json 2> $O/pr_step.err || exit 1
T=$(find $O/pr_step -name "*kernel_trace.csv" | head -1)
python scripts/step_launch_counts.py $T > $O/r04_train_step_one_step.txt 2>&1
python scripts/queue_balance.py $T > $O/r04_train_step_queues.txt 2>&1
cp $(find $O/pr_step -name "*kernel_stats.csv" | head -1) $O/r04_train_step_kernel_stats.csv
rm -rf $O/pr_step
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr_dec -- python3 scripts/bench_decode_r04.py > $O/pr_dec.log 2>&1 || exit 1
cp $(find $O/pr_dec -name "*kernel_stats.csv" | head -1) $O/r04_decode_turn_kernel_stats.csv
rm -rf $O/pr_dec
tail -3 $O/r04_attn_fwd_B64_kernel_stats.csv; head -12 $O/r04_train_step_one_step.txt; cat $O/r04_train_step_queues.txt
python -c "
import json; d=json.load(open('$O/r04_attn_fwd_B64_pmc.json')); print({k:v for k,v in d.items() if not isinstance(v,(dict,list))})"
