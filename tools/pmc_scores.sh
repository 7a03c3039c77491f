#!/bin/bash
# PMC passes for the scores kernel (run on the GPU box through gpurun).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { timeout 150 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "scores_kernel" --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1; }
run d "FETCH_SIZE"
run e "WRITE_SIZE"
run c "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run f "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
