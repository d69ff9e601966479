#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
echo "== nw4all: training tests"; NF_MI355X_LIB=$V/nw4all.so timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "benchmark_shape or pair or one_call or flat_parameters or whole_layer or full_fwd" 2>&1 | tail -3
for b in 4096 16384 32768 65536; do
echo "nw8 B=$b: $(timeout 300 python tools/train_bench.py --steps 6 --flat --batch $b 2>&1 | tail -1 | cut -c1-70)"
echo "nw4 B=$b: $(NF_MI355X_LIB=$V/nw4all.so timeout 300 python tools/train_bench.py --steps 6 --flat --batch $b 2>&1 | tail -1 | cut -c1-70)"
done
