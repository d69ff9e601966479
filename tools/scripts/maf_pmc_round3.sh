R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/maf_$n -- python $R/tools/config_bench.py 5 > $O/maf_$n.log 2>&1; echo "maf $n rc=$?"
done
cd $R
python tools/summarize_profiles.py r03_config5_maf_h --pmc $(find $O/maf_FETCH_SIZE $O/maf_WRITE_SIZE $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") --trace $(find $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "maf_inverse_h_kernel"
rm -rf $O
