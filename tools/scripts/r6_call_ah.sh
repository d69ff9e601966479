#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_hygiene.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python tools/batch_sweep.py --json gpurun_out/r06_batch_sweep.json 2>&1 | grep rows
