# rocprofv3 evidence for BASELINE configs[4]'s inverse pass at the round-5 tree: kernel stats of tools/maf_inverse_bench.py (the
# inverse kernels only: no training legs, no library-GEMM comparison in the profiled command) + counter passes, each counter set in
# its own run with --kernel-trace only.  Usage (on the GPU box): bash tools/scripts/r5_maf_profile.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5maf; mkdir -p $O
M="python $R/tools/maf_inverse_bench.py --reps 3"
cd /tmp && export TMPDIR=/tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $M > $O/stats.log 2>&1; echo "maf stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- $M > $O/pmc_$n.log 2>&1; echo "maf pmc $n rc=$?"
done
cd $R
python tools/summarize_profiles.py r05_config5_maf_inverse --stats $(find $O/stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "maf_inverse_h_kernel<2, true>" > $O/summ.log 2>&1
mkdir -p $R/gpurun_out/profiles_out; cp profiles/r05_config5_maf_inverse_* $R/gpurun_out/profiles_out/ 2>/dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/summ.log | head -20; head -6 profiles/r05_config5_maf_inverse_kernel_stats.csv | cut -c1-200
