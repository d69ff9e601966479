V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "resblock or one_call or benchmark_shape or whole_layer" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1
NF_MI355X_LIB=$V/bbnopipe.so timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1
done
timeout 120 python tools/resblock_probe.py 2>&1 | tail -4
