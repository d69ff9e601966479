#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "128_row_tiles or round6_switches_off or glow_training_step" 2>&1 | grep -v Warn | tail -4
for v in 0 1; do (NF_MADE_TR128=$v NF_AB=none timeout 300 python tools/glow_leaf_ab.py) 2> /dev/null | grep "^{" | cut -c1-160; done
