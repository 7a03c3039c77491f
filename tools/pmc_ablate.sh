#!/bin/bash
# MFMA pipe occupancy and sustained clock of the similarity-kernel variants of tools/scores_ablate (and its
# -DMDX_ABL_M32 build): one rocprofv3 --pmc pass per binary / mode, kernel-trace only (gpurun refuses --pmc with sys traces).
#   bash tools/pmc_ablate.sh     (through gpurun; writes gpurun_out/pmc_ablate/*.csv + summary.txt)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_ablate; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {  # name, env assignment or "", binary
  rm -rf /tmp/pa_$1
  if [ -n "$2" ]; then export $2; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/pa_$1 -- $3 > $OUT/$1.log 2>&1
  if [ -n "$2" ]; then unset ${2%%=*}; fi
  cp /tmp/pa_$1/*/*_counter_collection.csv $OUT/$1.csv 2>/dev/null
}
run base "" $R/tools/scores_ablate
run m32 "" $R/tools/scores_ablate_m32
run cw8 "CW8=1" $R/tools/scores_ablate
python3 - <<PY > $OUT/summary.txt
import csv, collections, glob, os
for f in sorted(glob.glob("$OUT/*.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print("==", os.path.basename(f))
    for k, c in d.items():
        if not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_VALU_MFMA_BUSY_CYCLES"): continue
        g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
        if g < 1e6: continue
        b = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        ms = sum(dur[k]) / len(dur[k]) / 1e6
        print("%-90s launches %3d  %.3f ms (counter pass)  clock %.2f GHz  MFMA pipe busy %.1f %%" % (k[:90], len(dur[k]), ms, g / ms / 1e6, 100 * b / 1024 / g))
PY
cat $OUT/summary.txt
