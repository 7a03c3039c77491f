#!/bin/bash
# per-kernel times of the ranking (rocprofv3 kernel trace of tools/quick_bench.py), summary to stdout
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; W=/tmp/prof_rank_$$; rm -rf $W
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $W -- python3 $R/tools/quick_bench.py "$@" > $W.log 2>&1
python3 - "$W" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "mdx::" in r["Name"] and ("sort" in r["Name"] or "msd" in r["Name"] or "os_" in r["Name"] or "scores" in r["Name"]):
        print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
