#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "in_place or solve_scratch or captures_into_one_graph or maf" 2>&1 | grep -v Warn | tail -4
timeout 300 python tools/maf_wgrad_pos_ab.py 2>&1 | grep -v Warn | tail -2
bash tools/scripts/r6_call_ar.sh
