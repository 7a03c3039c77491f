#!/bin/bash
# per-kernel times of the rOxford5k-sized step (70 x 4 993): rocprofv3 kernel trace of tools/quick_bench.py 4993
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; W=/tmp/prof_small_$$; rm -rf $W
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $W -- python3 $R/tools/quick_bench.py 4993 > $W.log 2>&1
python3 - "$W" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "mdx::" in r["Name"]:
        print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep -E "scores|rank_full" $W.log
