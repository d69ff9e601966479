#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | grep -v Warn | tail -6
