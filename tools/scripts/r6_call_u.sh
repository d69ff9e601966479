#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "side_stream or flat_parameters or pair or one_call or benchmark_shape" 2>&1 | tail -15
for i in 1 2; do
timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-200
done
timeout 300 python bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
