# Round-6 final evidence (last session): tools/scripts/evidence_round6.sh (full GPU suite, smoke, bench lines, kernel stats + counter passes)
# followed by the last session's own measurements -- kernel stats of config 5's density-direction step and of config 4's training step,
# the A/B tools of the round's last changes.  ONE gpurun call (python tools/stage_reference.py first).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ev
bash $R/tools/scripts/evidence_round6.sh
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mafd_stats -- python $R/tools/maf_density_profile.py --steps 5 > $O/mafd_stats.log 2>&1; echo "maf density stats rc=$?"
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/glowt_stats -- python $R/tools/glow_train_bench.py > $O/glowt_stats.log 2>&1; echo "glow train stats rc=$?"
cd $R
cp $(find $O/mafd_stats -name "*kernel_stats.csv" | head -1) profiles/r06_maf_density_step_kernel_stats.csv 2>/dev/null
cp $(find $O/glowt_stats -name "*kernel_stats.csv" | head -1) profiles/r06_config4_glow_train_kernel_stats.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
(timeout 300 python tools/maf_wgrad_pos_ab.py) 2> /dev/null | grep "^{" > profiles/r06_maf_wgrad_pos_ab.json
(timeout 300 python tools/maf_solve_fast_ab.py) 2> /dev/null | grep "^{" > profiles/r06_maf_solve_fast_ab.json
for k in weights_batched lazy_logdet leaf_async; do
  (NF_AB=$k timeout 400 python tools/glow_leaf_ab.py) 2> /dev/null | grep "^{" | sed "s/^{/{\"switch\": \"$k\", /" >> profiles/r06_glow_train_ab.jsonl
done
cp profiles/r06_* $R/gpurun_out/profiles_out/ 2>/dev/null
echo; cat profiles/r06_maf_wgrad_pos_ab.json; head -c 600 profiles/r06_maf_solve_fast_ab.json; echo; cat profiles/r06_glow_train_ab.jsonl | cut -c1-400
