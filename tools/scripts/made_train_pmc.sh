# rocprofv3 counter passes of MADE under autograd at BASELINE configs[4]'s layer (tools/made_train_bench.py --only): one summary per
# kernel (made_bwd_kernel, made_wgrad_kernel, made_fwd_kernel<2, 3>); every counter set in its own pass, --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4mt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/made_train_bench.py --only > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
cd $R
for k in made_bwd_kernel made_wgrad_kernel made_fwd_kernel; do
  python tools/summarize_profiles.py r04_made_train_$k --pmc $(find $O/pmc_* -name "*counter_collection.csv") --trace $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "$k" | tail -3
done
mkdir -p $R/gpurun_out/profiles_out; cp $R/profiles/r04_made_train_made_* $R/gpurun_out/profiles_out/
