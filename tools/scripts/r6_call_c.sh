# round 6, call C: final_bwd with two 4-wave workgroups per CU (variant fb4) vs the product build; the 16-byte reduction
mkdir -p gpurun_out/c
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for i in 1 2; do
timeout 200 python tools/final_bwd_probe.py 2>&1 | tail -1
NF_MI355X_LIB=$V/fb4.so timeout 200 python tools/final_bwd_probe.py 2>&1 | tail -1
done
NF_MI355X_LIB=$V/fb4.so timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "final_layer_backward_one_pass or one_call or benchmark_shape" > gpurun_out/c/pytest_fb4.log 2>&1; tail -3 gpurun_out/c/pytest_fb4.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "one_call or benchmark_shape or flat_parameters" > gpurun_out/c/pytest.log 2>&1; tail -3 gpurun_out/c/pytest.log | cut -c1-300
timeout 300 python tools/train_bench.py --steps 8 --flat > gpurun_out/c/train.log 2>&1; echo "product: $(tail -1 gpurun_out/c/train.log | cut -c1-200)"
NF_MI355X_LIB=$V/fb4.so timeout 300 python tools/train_bench.py --steps 8 --flat > gpurun_out/c/train_fb4.log 2>&1; echo "fb4: $(tail -1 gpurun_out/c/train_fb4.log | cut -c1-200)"
