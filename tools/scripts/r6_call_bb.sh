#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "128_row_tiles" 2>&1 | grep -v Warn | tail -4
