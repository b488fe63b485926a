# SQ counter passes of round 4: MFMA busy, LDS conflicts, wave wait shares per kernel for the roofline region (fused
# stage 1, stage 2), one eager TRAINING step (fused stage-1 training form, st1_pbwd, stage-2 backward) and decode turns (decstack).
# Counters only beside --kernel-trace (no other trace domain).  usage: bash scripts/prof_sq_r04.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
export BIST_SPLIT_GRAPH=0
rm -rf $O/sq_region $O/sq_step $O/sq_dec
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_region -- python3 scripts/prof_attn.py --B 64 --iters 3 > $O/sq_region.log 2>&1 || exit 1
python scripts/pmc_by_kernel.py $(find $O/sq_region -name "*counter_collection.csv" | head -1) > $O/r04_attn_fwd_B64_sq_counters.txt
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_step -- python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-decode --no-t128 --no-f32 --no-fed > $O/sq_step.json 2> $O/sq_step.err || exit 1
python scripts/pmc_by_kernel.py $(find $O/sq_step -name "*counter_collection.csv" | head -1) > $O/r04_train_step_sq_counters.txt
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_dec -- python3 scripts/bench_decode_r04.py > $O/sq_dec.log 2>&1 || exit 1
python scripts/pmc_by_kernel.py $(find $O/sq_dec -name "*counter_collection.csv" | head -1) > $O/r04_decode_sq_counters.txt
rm -rf $O/sq_region $O/sq_step $O/sq_dec
grep -h "st1_fused\|st1_pbwd\|st2_mfma\|decstack" $O/r04_attn_fwd_B64_sq_counters.txt $O/r04_train_step_sq_counters.txt $O/r04_decode_sq_counters.txt | cut -c1-400
