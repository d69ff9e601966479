#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "NF_MADE_TR128=$v"
  (NF_MADE_TR128=$v NF_AB=none timeout 300 python tools/glow_leaf_ab.py) 2> /dev/null | grep "^{" | cut -c1-200
done
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -x -q -m gpu -k "glow or convnet or made or resnet or wide" 2>&1 | tail -3
