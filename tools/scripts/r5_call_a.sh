mkdir -p gpurun_out/a
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "maf" > gpurun_out/a/pytest_maf.log 2>&1; tail -5 gpurun_out/a/pytest_maf.log
timeout 200 python tools/maf_inverse_bench.py --ablate > gpurun_out/a/maf_bench.log 2>&1; cat gpurun_out/a/maf_bench.log | cut -c1-400
timeout 200 python tools/cpu_reference.py --ref .refstage --where "gpu box" --rows 16384 --repeats 1 --threads 32,8,64 --quick --out gpurun_out/r05_cpu_reference_gpubox.json > gpurun_out/a/cpuref.log 2>&1; tail -2 gpurun_out/a/cpuref.log | cut -c1-600
NF_REFERENCE_PATH=.refstage timeout 200 python -m pytest tests/test_gpu_parity.py -k "reference_own_containers or reference_style_container" -v -rs > gpurun_out/r05_reference_containers_gpubox.log 2>&1; tail -4 gpurun_out/r05_reference_containers_gpubox.log
