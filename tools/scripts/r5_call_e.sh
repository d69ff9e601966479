mkdir -p gpurun_out/e
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "maf" > gpurun_out/e/pytest_maf_train.log 2>&1; tail -12 gpurun_out/e/pytest_maf_train.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "maf or spline_debug" > gpurun_out/e/pytest_maf.log 2>&1; tail -3 gpurun_out/e/pytest_maf.log | cut -c1-300
timeout 300 python tools/maf_density_train_bench.py > gpurun_out/e/density.log 2>&1; tail -2 gpurun_out/e/density.log | cut -c1-600
