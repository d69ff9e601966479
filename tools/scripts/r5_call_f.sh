mkdir -p gpurun_out/f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/f/pytest_gpu.log 2>&1; tail -4 gpurun_out/f/pytest_gpu.log | cut -c1-300
timeout 300 python tools/wide_bench.py --json gpurun_out/f/wide_bench.json > gpurun_out/f/wide.log 2>&1; cat gpurun_out/f/wide.log | grep "D =" | cut -c1-200
NF_MI355X_LIB=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants/epi_scalar.so timeout 300 python tools/wide_bench.py --json gpurun_out/f/wide_bench_scalar_epilogue.json > gpurun_out/f/wide_scalar.log 2>&1; cat gpurun_out/f/wide_scalar.log | grep "D =" | cut -c1-200
timeout 300 python tools/train_bench.py --steps 5 --fused-adam > gpurun_out/f/train.log 2>&1; tail -3 gpurun_out/f/train.log | cut -c1-400
timeout 200 python tools/arnsf_density_bench.py > gpurun_out/f/arnsf.log 2>&1; tail -3 gpurun_out/f/arnsf.log | cut -c1-300
