#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "ragged or side_stream or flat_parameters" 2>&1 | tail -12
for b in 65537 10000 10048; do
echo "B=$b: $(timeout 300 python tools/train_bench.py --steps 5 --flat --batch $b 2>&1 | tail -1 | cut -c1-90)"
done
