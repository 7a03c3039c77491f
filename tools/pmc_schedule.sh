#!/bin/bash
# SQ counter pass of the exact similarity kernel in its two consumer schedules (MDX_SCORES_PIPE=0 / 1), same box, same workload
# (bench.py --profile: N = 1 004 993, Q = 70, D = 2048, gaussian unit rows), plus the in-process A/B timing on real rows and on
# all-zero operands (tools/scores_pipe_probe.py).  Summarised into profiles/r05_scores_schedule.md by tools/summarize_schedule.py.
#   bash tools/pmc_schedule.sh        (on the GPU box, through gpurun)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/schedule_r05; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES"
for p in 0 1 0 1; do
  W=/tmp/sched_${p}_$RANDOM; rm -rf $W
  MDX_SCORES_PIPE=$p timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "scores_lc" --output-format csv -d $W -- python3 $R/bench.py --steps 5 --warmup 2 --profile > $OUT/pmc_pipe$p.log 2>&1
  # (two runs per schedule go into one file: the CSV header only once)
  if [ -f $OUT/pmc_pipe$p.csv ]; then tail -q -n +2 $W/*/*_counter_collection.csv >> $OUT/pmc_pipe$p.csv; else cat $W/*/*_counter_collection.csv > $OUT/pmc_pipe$p.csv; fi
done
for i in 1 2 3; do timeout 300 python3 $R/tools/scores_pipe_probe.py 3 >> $OUT/ab.log 2>&1; done
grep round $OUT/ab.log | tail -20
