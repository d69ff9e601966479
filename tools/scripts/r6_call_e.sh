mkdir -p gpurun_out/e
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "pair_training or benchmark_shape or flat_parameters" > gpurun_out/e/pytest.log 2>&1; tail -2 gpurun_out/e/pytest.log | cut -c1-200
for i in 1 2; do
timeout 300 python tools/train_bench.py --steps 8 --flat > gpurun_out/e/train.log 2>&1; echo "pair: $(tail -1 gpurun_out/e/train.log | cut -c1-200)"
timeout 300 python tools/train_bench.py --steps 8 --flat --no-pair > gpurun_out/e/train_nopair.log 2>&1; echo "no pair: $(tail -1 gpurun_out/e/train_nopair.log | cut -c1-200)"
done
