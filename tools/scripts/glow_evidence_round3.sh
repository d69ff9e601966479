# Round-3 Glow (config 4) evidence, one gpurun call: kernel trace of config_bench 4 -> per-block times inside the level chains,
# the phase trace of workgroup 0 (debug build, tools/build_variant.py trace "-DNF_GL_TRACE" glow_conv.hip beforehand),
# and the counter passes of the three kernels.  Outputs under gpurun_out/r3g.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/config_bench.py 4 > $O/kt.log 2>&1; echo "trace rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/config_bench.py 4 > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
cd $R
python tools/glow_level_chains.py $(find $O/kt -name "*kernel_trace.csv" | head -1) --json $O/r03_config4_glow_level_chains.json > /dev/null
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r03_config4_glow_kernel_stats.csv
for k in glow_convnet_kernel glow_convnet_small_kernel glow_convnet_tiny_kernel; do
  python tools/summarize_profiles.py r03_config4_$k --pmc $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
    --trace $(find $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "$k" > /dev/null 2>&1
  cp profiles/r03_config4_${k}_pmc.json $O/ 2>/dev/null
done
if [ -f normalizing-flows_amd/lib/variants/trace.so ]; then
  echo "# NF_MI355X_LIB=normalizing-flows_amd/lib/variants/trace.so python tools/glow_trace.py   (1x MI355X, config 4, B = 256, density direction)" > $O/r03_glow_phase_trace.txt
  echo "# per GlowBlock inside a 32-block level chain, workgroup 0, mean over the 32 blocks (us)" >> $O/r03_glow_phase_trace.txt
  NF_MI355X_LIB=normalizing-flows_amd/lib/variants/trace.so python tools/glow_trace.py 2>/dev/null | tail -3 >> $O/r03_glow_phase_trace.txt
fi
python tools/config_bench.py 4 2>/dev/null | tail -1 > $O/config4.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/r03_config4_glow_level_chains.json | grep "per_block\|frac\|level\""; cat $O/r03_glow_phase_trace.txt; cat $O/config4.txt
