# The roofline region's evidence alone (run on the GPU box from the repo root): kernel trace + stats, then the two HBM counter passes
# (separate --pmc runs beside --kernel-trace only).  Needed whenever a source of the region's kernels changes: bench.py keys the committed
# counters by the sha256 of those sources.  usage: bash scripts/prof_region_r04.sh  ->  gpurun_out/r04_attn_fwd_B64_*
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
export BIST_SPLIT_TIMEOUT_MS=5000
rm -rf $O/pr_region $O/pr_fetch $O/pr_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pr_region -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_region.log 2>&1 || exit 1
T=$(find $O/pr_region -name "*kernel_trace.csv" | head -1)
python scripts/region_kernels.py $T 5 > $O/r04_attn_fwd_B64_kernel_stats.csv
cp $(find $O/pr_region -name "*kernel_stats.csv" | head -1) $O/r04_attn_fwd_B64_rocprofv3_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pr_fetch -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pr_write -- python3 scripts/prof_attn.py --B 64 --iters 5 > $O/pr_write.log 2>&1 || exit 1
python scripts/pmc_region.py $(find $O/pr_fetch -name "*counter_collection.csv" | head -1) $(find $O/pr_write -name "*counter_collection.csv" | head -1) 64 32 5 > $O/r04_attn_fwd_B64_pmc.json
rm -rf $O/pr_region $O/pr_fetch $O/pr_write
cat $O/r04_attn_fwd_B64_kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$O/r04_attn_fwd_B64_pmc.json')); print({k:v for k,v in d.items() if not isinstance(v,(dict,list))})"
