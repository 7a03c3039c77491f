#!/bin/bash
# Why does the two-stream form gain nothing?  Kernel trace of tools/overlap_run.py (serial + piped arrangements of the headline
# kernels) with the shipped similarity kernel (3-stage ring: 2 x 78 KiB of a CU's 160 KiB LDS) and with the probe-only two-stage
# ring (2 x 52 KiB: a 44-KiB sort workgroup fits beside them).  HISTORICAL: the MDX_SCORES_NSTAGE=2 switch exists in the library of
# commit 47a9fe2 only (`git checkout 47a9fe2 -- mdir_amd/csrc` in a scratch tree to reproduce profiles/r05_overlap.md); with the
# current library this script traces the shipped 3-stage kernel once.  Summarised by tools/summarize_overlap.py.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/overlap_r05; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for st in 3; do
  W=/tmp/ovl_$st; rm -rf $W
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $W -- python3 $R/tools/overlap_run.py 10 > $OUT/run_nstage$st.log 2>&1
  python3 - $W $OUT/trace_nstage$st.csv <<'PY'
import csv, glob, sys
src = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
with open(sys.argv[2], "w") as f:
    f.write("kernel,queue,start_ns,end_ns\n")
    for r in csv.DictReader(open(src)):
        k = r["Kernel_Name"]
        if "mdx::" in k:
            f.write("%s,%s,%s,%s\n" % (k.split("(")[0].replace("void ", "").replace(",", ";"), r["Queue_Id"], r["Start_Timestamp"], r["End_Timestamp"]))
PY
  # the same without the profiler (its serialisation of queues, if any, would hide the answer)
  timeout 300 python3 $R/tools/overlap_run.py 20 > $OUT/plain_nstage$st.log 2>&1
done
grep -h "ms/step\|identical" $OUT/plain_nstage*.log $OUT/run_nstage*.log
