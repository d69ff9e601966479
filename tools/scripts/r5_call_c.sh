mkdir -p gpurun_out/c
timeout 500 python tools/maf_ablate5.py run base lb8 hnw2 hnw8 > gpurun_out/c/ablate.jsonl 2>&1; cat gpurun_out/c/ablate.jsonl | cut -c1-200
bash tools/scripts/r5_maf_profile.sh > gpurun_out/c/maf_profile.log 2>&1; tail -30 gpurun_out/c/maf_profile.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "eight_ranks or two_ranks" > gpurun_out/c/pytest_contract.log 2>&1; tail -3 gpurun_out/c/pytest_contract.log | cut -c1-300
timeout 120 python tools/config_bench.py 1 > gpurun_out/c/c1.log 2>&1; tail -2 gpurun_out/c/c1.log | cut -c1-400
