# Round-3 evidence at the final tree, one gpurun call: full GPU suite, smoke, bench line, rocprofv3 kernel stats and the counter
# passes (FETCH_SIZE, WRITE_SIZE, MFMA-busy: each in its own pass with --kernel-trace only) of the bench chain, the training step
# and the MAF inverse (config 5).  Outputs under gpurun_out/r3ev; tools/scripts/evidence_round3_summarise.sh condenses them into profiles/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3ev
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -8 > $O/pytest_gpu.log     # (bounded: one call of this script once sat 40 minutes in this line on a bad box)
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
T="python $R/tools/train_bench.py --steps 3 --fused-adam"
M="python $R/tools/config_bench.py 5"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_stats.log 2>&1; echo "bench stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python $R/tools/train_bench.py --steps 5 --fused-adam > $O/train_stats.log 2>&1; echo "train stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$n -- $B > $O/bench_$n.log 2>&1; echo "bench $n rc=$?"
done
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/train_$n -- $T > $O/train_$n.log 2>&1; echo "train $n rc=$?"
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/maf_$n -- $M > $O/maf_$n.log 2>&1; echo "maf $n rc=$?"
done
# keep the merge small: the per-dispatch counter tables are condensed on the box
cd $R
python tools/summarize_profiles.py r03_bench_chain --stats $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "rqs_fused_kernel<0, true" > $O/summ_bench.log 2>&1
python tools/summarize_profiles.py r03_config5_maf_h --pmc $(find $O/maf_FETCH_SIZE $O/maf_WRITE_SIZE $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "maf_inverse_h_kernel" > $O/summ_maf.log 2>&1
python tools/pmc_summary.py $O train "python tools/train_bench.py --steps 3 --fused-adam" > $O/r03_train_step_pmc.json 2> $O/summ_train.log
cp $(find $O/train_stats -name "*kernel_stats.csv" | head -1) $O/r03_train_step_kernel_stats.csv
mkdir -p $O/profiles_out; cp profiles/r03_bench_chain* profiles/r03_config5_maf_h* $O/profiles_out/ 2>/dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench_line.json | head -c 600
