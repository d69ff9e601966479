#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -x -q -m gpu -k "glow or inv1x1 or captures_into_one_graph or side_stream" 2>&1 | grep -v Warn | tail -6
NF_AB=weights_batched timeout 600 python tools/glow_leaf_ab.py 2>&1 | grep -v Warn | tail -2
