#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -x -q -m gpu -k "maf or captures_into_one_graph or autoregressive" 2>&1 | grep -v Warn | tail -4
timeout 300 python tools/maf_wgrad_pos_ab.py 2>&1 | grep -v Warn | tail -1
(timeout 600 python tools/config_bench.py 5) 2> /dev/null | grep "^{\|^config" | cut -c1-900
