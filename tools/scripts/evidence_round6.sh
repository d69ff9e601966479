# Round-6 evidence at the final tree, ONE gpurun call (python tools/stage_reference.py first): full GPU suite incl. the real-container
# drop-in test against the staged reference, smoke, the contract line (default) and the --train line, rocprofv3 kernel stats + counter
# passes (each counter set in its own pass, --kernel-trace only) of the bench chain, the training step, config 5's inverse pass and
# config 4 (per kernel family with --kernel-include-regex: the whole-model counter pass crashed rocprofv3 in round 5), bench tables.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ev
mkdir -p $O $R/gpurun_out/profiles_out
cd $R
NF_REFERENCE_PATH=$R/.refstage timeout 1200 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12 > $O/pytest_gpu.log
NF_REFERENCE_PATH=$R/.refstage timeout 300 python -m pytest tests/test_gpu_parity.py -k "reference_own_containers or reference_style_container" -v 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" | cut -c1-200 > profiles/r06_reference_containers_gpubox.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 300 python bench.py --train --steps 10 --warmup 3 > $O/bench_train_line.json 2> $O/bench_train.err
timeout 400 python tools/wide_bench.py --json $R/profiles/r06_wide_bench.json > $O/wide_bench.log 2>&1
timeout 400 python tools/kernel_bench.py --json $R/profiles/r06_kernel_bench.json > $O/kernel_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
T="python $R/tools/train_bench.py --steps 3 --flat"
M="python $R/tools/maf_inverse_bench.py --reps 3"
G="python $R/tools/config_bench.py 4"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_stats.log 2>&1; echo "bench stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python $R/tools/train_bench.py --steps 5 --flat > $O/train_stats.log 2>&1; echo "train stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/glow_stats -- $G > $O/glow_stats.log 2>&1; echo "glow stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/maf_stats -- $M > $O/maf_stats.log 2>&1; echo "maf stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$n -- $B > $O/bench_$n.log 2>&1; echo "bench $n rc=$?"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/train_$n -- $T > $O/train_$n.log 2>&1; echo "train $n rc=$?"
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/maf_$n -- $M > $O/maf_$n.log 2>&1; echo "maf $n rc=$?"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "glow_convnet" --output-format csv -d $O/glow_$n -- $G > $O/glow_$n.log 2>&1; echo "glow $n rc=$?"
done
cd $R
python tools/summarize_profiles.py r06_bench_chain --stats $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "rqs_fused_kernel<0, true" > $O/summ_bench.log 2>&1
python tools/summarize_profiles.py r06_config5_maf_inverse --stats $(find $O/maf_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/maf_FETCH_SIZE $O/maf_WRITE_SIZE $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/maf_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "maf_inverse_h_kernel<2, true" > $O/summ_maf.log 2>&1
python tools/pmc_summary.py $O train "python tools/train_bench.py --steps 3 --flat" > profiles/r06_train_step_pmc.json 2> $O/summ_train.log
python tools/pmc_summary.py $O glow "python tools/config_bench.py 4 (rocprofv3 --kernel-include-regex glow_convnet)" > profiles/r06_config4_glow_pmc.json 2> $O/summ_glow.log
cp $(find $O/glow_stats -name "*kernel_stats.csv" | head -1) profiles/r06_config4_glow_kernel_stats.csv 2>/dev/null
python tools/glow_level_chains.py $(find $O/glow_stats -name "*kernel_trace.csv" | head -1) --json profiles/r06_config4_glow_level_chains.json > /dev/null 2>&1
cp $(find $O/train_stats -name "*kernel_stats.csv" | head -1) profiles/r06_train_step_kernel_stats.csv 2>/dev/null
head -c 40000 $O/bench_line.json | tail -1 > profiles/r06_bench_line.json
tail -1 $O/bench_train_line.json > profiles/r06_bench_train_line.json
tail -12 $O/pytest_gpu.log > profiles/r06_pytest_gpu.log; tail -5 $O/smoke.log >> profiles/r06_pytest_gpu.log
python tools/kernel_resources.py > profiles/r06_kernel_resources.txt 2>/dev/null
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
(python tools/config_bench.py 5) 2> /dev/null | grep "^{\|^config" > profiles/r06_maf.jsonl
cp profiles/r06_* $R/gpurun_out/profiles_out/ 2>/dev/null
tail -3 $O/pytest_gpu.log | cut -c1-200; cat profiles/r06_reference_containers_gpubox.log | tail -3; tail -2 $O/smoke.log | cut -c1-200; head -c 500 profiles/r06_bench_line.json; echo; head -c 600 profiles/r06_bench_train_line.json; echo; tail -8 $O/summ_bench.log; head -c 700 profiles/r06_config4_glow_pmc.json
