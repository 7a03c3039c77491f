#!/bin/bash
# Round profile on the GPU box (through gpurun): kernel trace + stats of the HEADLINE LOOP ALONE (bench.py --profile: no
# two-stream leg, no side legs, no extraction -- so that the per-kernel averages add up to the step), a second stats run of
# the extraction leg, then HBM-traffic / SQ PMC passes in their own runs (MI355X_MICROARCH.md, HBM section).
# Only small summaries are kept (gpurun_out is capped at 64 MiB).
#   bash tools/profile_round.sh r04
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/profile_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $R/bench.py --profile > $OUT/stats_bench.log 2>&1
cp $W/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
head -1 $W/stats/*/*_kernel_trace.csv > $OUT/kernel_trace_mdx.csv; grep "mdx::" $W/stats/*/*_kernel_trace.csv | head -4000 >> $OUT/kernel_trace_mdx.csv
if [ "${EXTRACT:-1}" = "1" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats_x -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-pipelined --extract-images 16 > $OUT/stats_extract.log 2>&1
  cp $W/stats_x/*/*_kernel_stats.csv $OUT/kernel_stats_extract.csv
fi
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "mdx::" --output-format csv -d $W/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --profile ${PMC_ARGS:-} > $OUT/pmc_$n.log 2>&1
  cp $W/pmc_$n/*/*_counter_collection.csv $OUT/pmc_$n.csv
done
# the labelled split-precision modes (MDX_F32_SPLIT3, MDX_F32_SPLIT2) on the same shard: their own counter passes over tools/split_bench.py
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "scores_split" --output-format csv -d $W/pmc3_$n -- python3 $R/tools/split_bench.py 1004993 3 > $OUT/pmc_split3_$n.log 2>&1
  cp $W/pmc3_$n/*/*_counter_collection.csv $OUT/pmc_split3_$n.csv
done
timeout 200 python3 $R/tools/split_bench.py > $OUT/split_bench.log 2>&1
tail -1 $OUT/stats_bench.log | cut -c1-300
du -sh $OUT
