#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "nsf_wide_one_launch" 2>&1 | grep -E "^E|assert|FAILED|passed|failed" | head -30
