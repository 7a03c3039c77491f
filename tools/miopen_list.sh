#!/bin/bash
# profiles/r05_miopen.md: the 16 sizes x 3 scales of the extraction list under FAST (package default) and under a full find
# persisted to a user find-db, same box, order fast / find / fast / find-again (the last run reads the find-db the second wrote).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/miopen_r05; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
D=/tmp/miopen_userdb; rm -rf $D; mkdir -p $D/db $D/cache
export MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache MIOPEN_LOG_LEVEL=1
MIOPEN_FIND_MODE=2 timeout 900 python3 $R/tools/miopen_list_probe.py fast_1 2>/dev/null | grep '^{' > $OUT/fast_1.jsonl
MIOPEN_FIND_MODE=1 timeout 2400 python3 $R/tools/miopen_list_probe.py find_1 benchmark 2>/dev/null | grep '^{' > $OUT/find_1.jsonl
MIOPEN_FIND_MODE=2 timeout 900 python3 $R/tools/miopen_list_probe.py fast_2 2>/dev/null | grep '^{' > $OUT/fast_2.jsonl
MIOPEN_FIND_MODE=1 timeout 2400 python3 $R/tools/miopen_list_probe.py find_2 benchmark 2>/dev/null | grep '^{' > $OUT/find_2.jsonl
tail -qn1 $OUT/*.jsonl; du -sh $D/db
