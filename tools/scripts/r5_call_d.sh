mkdir -p gpurun_out/d
bash tools/scripts/r5_epilogue_ab.sh > gpurun_out/d/epi.log 2>&1; tail -45 gpurun_out/d/epi.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/d/pytest_gpu.log 2>&1; tail -6 gpurun_out/d/pytest_gpu.log | cut -c1-300
