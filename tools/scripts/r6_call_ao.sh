#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "side_stream or captures_into_one_graph or glow" 2>&1 | tail -15
timeout 600 python tools/glow_leaf_ab.py 2>&1 | grep -v Warn | tail -3
