# Round-5 evidence, the part that changed after tools/scripts/evidence_round5.sh ran (MAF implicit backward on the kernels' own scratch,
# Glow small kernels, AR implicit inverse): full GPU suite with the staged reference, smoke, the contract line, the config-5 tables.
# The rocprofv3 counter passes, the reference's CPU leg and the wide / kernel bench tables of the first script stand.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5tail
mkdir -p $O $R/gpurun_out/profiles_out
cd $R
NF_REFERENCE_PATH=$R/.refstage timeout 900 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12 > $O/pytest_gpu.log
NF_REFERENCE_PATH=$R/.refstage timeout 300 python -m pytest tests/test_gpu_parity.py -k "reference_own_containers or reference_style_container" -v 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" | cut -c1-200 > profiles/r05_reference_containers_gpubox.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
head -c 30000 $O/bench_line.json | tail -1 > profiles/r05_bench_line.json
tail -12 $O/pytest_gpu.log > profiles/r05_pytest_gpu.log; tail -4 $O/smoke.log >> profiles/r05_pytest_gpu.log
(python tools/maf_inverse_bench.py --ablate; python tools/maf_density_train_bench.py; python tools/made_train_bench.py; python tools/maf_train_bench.py; python tools/config_bench.py 5) 2> /dev/null | grep "^{\|^config" > profiles/r05_maf.jsonl
timeout 300 python tools/ar_implicit_bench.py --out profiles/r05_ar_implicit.json > $O/ar_implicit.log 2>&1
timeout 200 python tools/train_launch_audit.py --model glow --out profiles/r05_glow_launch_audit.json > $O/glow_audit.log 2>&1
cp profiles/r05_* $R/gpurun_out/profiles_out/ 2>/dev/null
tail -3 $O/pytest_gpu.log | cut -c1-200; cat profiles/r05_reference_containers_gpubox.log | tail -3; tail -2 $O/smoke.log | cut -c1-200; head -c 500 profiles/r05_bench_line.json; echo; grep "^config" profiles/r05_maf.jsonl | cut -c1-500
