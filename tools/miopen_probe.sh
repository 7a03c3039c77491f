#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
# every variant starts from EMPTY MIOpen user databases / kernel caches (a fresh box), or the first variant pays for all
run() { local tag=$1; shift; D=/tmp/miopen_$tag; rm -rf $D; mkdir -p $D/db $D/cache
        env MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache "$@" python3 $R/tools/miopen_probe.py 683 1024 2>&1 | grep call | tr '\n' ' '; echo " <- $tag"; }
run default MIOPEN_FIND_MODE=5
run nonaive MIOPEN_FIND_MODE=5 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
run findmode1 MIOPEN_FIND_MODE=1
run findmode2 MIOPEN_FIND_MODE=2
run findmode3 MIOPEN_FIND_MODE=3
run findmode4 MIOPEN_FIND_MODE=4
