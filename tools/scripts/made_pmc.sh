# rocprofv3 kernel statistics + counter passes of the one-launch MADE forward at BASELINE configs[4]'s layer shape
# (tools/made_bench.py); every counter set in its own pass, --kernel-trace only (no sys/hip trace with --pmc).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
TAG=${1:-r04_made_fwd}
cd /tmp && export TMPDIR=/tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/made_bench.py > $O/stats.log 2>&1; echo "stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/made_bench.py > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
cd $R
python tools/summarize_profiles.py $TAG --stats $(find $O/stats -name "*kernel_stats.csv" | head -1) --pmc $(find $O/pmc_* -name "*counter_collection.csv") --trace $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "made_fwd_kernel"
mkdir -p $R/gpurun_out/profiles_out; cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_out/
tail -2 $O/stats.log | cut -c1-300
