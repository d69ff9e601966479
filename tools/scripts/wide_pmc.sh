# counter passes of nf_nsf_wide on one wide shape (tools/wide_bench.py D H): kernel trace + SQ counters, one set per pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O
D=${1:-128}; H=${2:-128}; TAG=${3:-r04_nsf_wide_d${D}_h${H}}
cd /tmp && export TMPDIR=/tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/wide_bench.py --only $D $H > $O/stats.log 2>&1; echo "stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/wide_bench.py --only $D $H > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
cd $R
python tools/summarize_profiles.py $TAG --stats $(find $O/stats -name "*kernel_stats.csv" | head -1) --pmc $(find $O/pmc_* -name "*counter_collection.csv") --trace $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "nsf_wide_kernel"
mkdir -p $R/gpurun_out/profiles_out; cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_out/
rm -rf $O
