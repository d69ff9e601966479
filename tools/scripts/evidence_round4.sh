# Round-4 evidence at the final tree, one gpurun call: full GPU suite, smoke, the contract line, rocprofv3 kernel stats + counter passes
# (each counter set in its own pass, --kernel-trace only) of the bench chain, config 5 (both directions), the one-launch MADE forward
# and nf_nsf_wide, the wide / kernel bench tables and the training step's kernel stats.  Condensed summaries land in profiles/ (and a
# copy under gpurun_out/profiles_out, which is what travels back).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4ev
mkdir -p $O $R/gpurun_out/profiles_out
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 400 python tools/wide_bench.py --json $R/profiles/r04_wide_bench.json > $O/wide_bench.log 2>&1
timeout 400 python tools/kernel_bench.py --json $R/profiles/r04_kernel_bench.json > $O/kernel_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
M="python $R/tools/config_bench.py 5"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_stats.log 2>&1; echo "bench stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/maf_stats -- $M > $O/maf_stats.log 2>&1; echo "maf stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python $R/tools/train_bench.py --steps 5 --fused-adam > $O/train_stats.log 2>&1; echo "train stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$n -- $B > $O/bench_$n.log 2>&1; echo "bench $n rc=$?"
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/maf_$n -- $M > $O/maf_$n.log 2>&1; echo "maf $n rc=$?"
done
cd $R
python tools/summarize_profiles.py r04_bench_chain --stats $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "rqs_fused_kernel<0, true" > $O/summ_bench.log 2>&1
python tools/summarize_profiles.py r04_config5_maf --stats $(find $O/maf_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/maf_FETCH_SIZE $O/maf_WRITE_SIZE $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "made_fwd_kernel" > $O/summ_maf.log 2>&1
mv profiles/r04_config5_maf_pmc.json profiles/r04_config5_maf_forward_pmc.json 2>/dev/null
python tools/summarize_profiles.py r04_config5_maf_inverse --pmc $(find $O/maf_FETCH_SIZE $O/maf_WRITE_SIZE $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "maf_inverse_h_kernel" >> $O/summ_maf.log 2>&1
cp $(find $O/train_stats -name "*kernel_stats.csv" | head -1) profiles/r04_train_step_kernel_stats.csv 2>/dev/null
head -c 20000 $O/bench_line.json | tail -1 > profiles/r04_bench_line.json
tail -8 $O/pytest_gpu.log > profiles/r04_pytest_gpu.log; tail -4 $O/smoke.log >> profiles/r04_pytest_gpu.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
bash tools/scripts/made_pmc.sh r04_made_fwd > $O/made_pmc.log 2>&1
bash tools/scripts/wide_pmc.sh 64 256 > $O/wide_pmc_64_256.log 2>&1
bash tools/scripts/wide_pmc.sh 128 128 > $O/wide_pmc_128_128.log 2>&1
# MADE under autograd (csrc/made_bwd.hip): per-kernel split at config 5's layer, hand-written vs library path, the 10-layer step
(cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/made_train_stats -- python $R/tools/made_train_bench.py --only > $O/made_train_stats.log 2>&1)
cp $(find $O/made_train_stats -name "*kernel_stats.csv" | head -1) profiles/r04_made_train_kernel_stats.csv 2>/dev/null
(python tools/made_train_bench.py; python tools/made_train_bench.py --mult 23 --batch 16384; python tools/maf_train_bench.py; python tools/wide_train_bench.py; python tools/glow_train_bench.py; python tools/maf_density_train_bench.py) 2> /dev/null | grep "^{" > profiles/r04_made_train.jsonl
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/glow_train_stats -- python $R/tools/glow_train_bench.py > $O/glow_train_stats.log 2>&1)
cp $(find $O/glow_train_stats -name "*kernel_stats.csv" | head -1) profiles/r04_glow_train_kernel_stats.csv 2>/dev/null
cp profiles/r04_* $R/gpurun_out/profiles_out/ 2>/dev/null
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -c 400 profiles/r04_bench_line.json
