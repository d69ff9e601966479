#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "side_stream" 2>&1 | grep -v Warn | grep -B30 "short test summary" | head -60
