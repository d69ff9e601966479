# Round-5 evidence at the final tree, ONE gpurun call (stage the reference first: python tools/stage_reference.py): the full GPU suite
# incl. the real-container drop-in test against the staged reference, smoke, the contract line, the reference's CPU leg on the box's
# host cores, rocprofv3 kernel stats + counter passes (each counter set in its own pass, --kernel-trace only) of the bench chain,
# config 5's inverse pass and config 4, the wide / kernel bench tables, the training steps.  Condensed summaries land in profiles/
# (copied to gpurun_out/profiles_out, which is what travels back).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5ev
mkdir -p $O $R/gpurun_out/profiles_out
cd $R
NF_REFERENCE_PATH=$R/.refstage timeout 900 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12 > $O/pytest_gpu.log
NF_REFERENCE_PATH=$R/.refstage timeout 300 python -m pytest tests/test_gpu_parity.py -k "reference_own_containers or reference_style_container" -v 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" | cut -c1-200 > profiles/r05_reference_containers_gpubox.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
# the reference itself on this box's host cores: the full benchmark batch, thread sweep, both directions
if [ -d $R/.refstage/normflows ]; then
  timeout 420 python tools/cpu_reference.py --ref .refstage --where "gpu box" --rows 65536 --repeats 2 --threads 8,16 --out profiles/r05_cpu_reference_gpubox.json > $O/cpuref.log 2>&1
fi
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 400 python tools/wide_bench.py --json $R/profiles/r05_wide_bench.json > $O/wide_bench.log 2>&1
timeout 400 python tools/kernel_bench.py --json $R/profiles/r05_kernel_bench.json > $O/kernel_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_stats.log 2>&1; echo "bench stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python $R/tools/train_bench.py --steps 5 --fused-adam > $O/train_stats.log 2>&1; echo "train stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/glow_stats -- python $R/tools/config_bench.py 4 > $O/glow_stats.log 2>&1; echo "glow stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mafkld_stats -- python $R/tools/maf_density_train_bench.py > $O/mafkld_stats.log 2>&1; echo "maf density stats rc=$?"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/bench_$n -- $B > $O/bench_$n.log 2>&1; echo "bench $n rc=$?"
  # (the counter passes of tools/config_bench.py 4 crashed rocprofv3 twice this round, rc 139: not repeated; the r04 Glow PMC file stands)
done
cd $R
python tools/summarize_profiles.py r05_bench_chain --stats $(find $O/bench_stats -name "*kernel_stats.csv" | head -1) \
  --pmc $(find $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") \
  --trace $(find $O/bench_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1) --kernel "rqs_fused_kernel<0, true" > $O/summ_bench.log 2>&1
cp $(find $O/glow_stats -name "*kernel_stats.csv" | head -1) profiles/r05_config4_glow_kernel_stats.csv 2>/dev/null
python tools/glow_level_chains.py $(find $O/glow_stats -name "*kernel_trace.csv" | head -1) --json profiles/r05_config4_glow_level_chains.json > /dev/null 2>&1
cp $(find $O/train_stats -name "*kernel_stats.csv" | head -1) profiles/r05_train_step_kernel_stats.csv 2>/dev/null
cp $(find $O/mafkld_stats -name "*kernel_stats.csv" | head -1) profiles/r05_maf_density_train_kernel_stats.csv 2>/dev/null
head -c 30000 $O/bench_line.json | tail -1 > profiles/r05_bench_line.json
tail -12 $O/pytest_gpu.log > profiles/r05_pytest_gpu.log; tail -4 $O/smoke.log >> profiles/r05_pytest_gpu.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
bash tools/scripts/r5_maf_profile.sh > $O/maf_profile.log 2>&1
(python tools/maf_inverse_bench.py --ablate; python tools/maf_density_train_bench.py; python tools/made_train_bench.py; python tools/maf_train_bench.py; python tools/config_bench.py 5) 2> /dev/null | grep "^{\|^config" > profiles/r05_maf.jsonl
timeout 300 python tools/ar_implicit_bench.py --out profiles/r05_ar_implicit.json > $O/ar_implicit.log 2>&1
timeout 200 python tools/train_launch_audit.py --out profiles/r05_train_launch_audit.json > $O/train_audit.log 2>&1
cp profiles/r05_* $R/gpurun_out/profiles_out/ 2>/dev/null
tail -3 $O/pytest_gpu.log | cut -c1-200; cat profiles/r05_reference_containers_gpubox.log | tail -4; tail -2 $O/smoke.log | cut -c1-200; head -c 600 profiles/r05_bench_line.json; echo; tail -2 $O/cpuref.log | cut -c1-400; tail -12 $O/summ_bench.log
