R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gram_prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 5 plain; do
timeout 50 $R/tools/gram_ablate_bin_$v
done
for v in 0 5; do
rm -rf /tmp/gp_$v
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-include-regex "gemm_f64" --output-format csv -d /tmp/gp_$v -- $R/tools/gram_ablate_bin_$v > $OUT/pmc_$v.log 2>&1
cp /tmp/gp_$v/*/*_counter_collection.csv $OUT/pmc_abl_$v.csv
cp /tmp/gp_$v/*/*_kernel_trace.csv $OUT/trace_abl_$v.csv
done
