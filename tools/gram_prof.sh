#!/bin/bash
# PMC pass over the f64 whitening-learning GEMMs (through gpurun): sustained clock and matrix-pipe occupancy of the product
# build and of the "no loads inside the loop" build of tools/gram_ablate.hip.  Build first, here:
#   for a in 0 5; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMDX_GRAM_ABL=$a -I mdir_amd/csrc -I include tools/gram_ablate.hip -o tools/gram_ablate_bin_$a; done
#   gpurun -- 'bash tools/gram_prof.sh'
# Then per kernel: clock = GRBM_GUI_ACTIVE / 8 / duration, occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gram_prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 5; do
  timeout 50 $R/tools/gram_ablate_bin_$v
  rm -rf /tmp/gp_$v
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-include-regex "gemm_f64" --output-format csv -d /tmp/gp_$v -- $R/tools/gram_ablate_bin_$v > $OUT/pmc_$v.log 2>&1
  cp /tmp/gp_$v/*/*_counter_collection.csv $OUT/pmc_abl_$v.csv
  cp /tmp/gp_$v/*/*_kernel_trace.csv $OUT/trace_abl_$v.csv
done
