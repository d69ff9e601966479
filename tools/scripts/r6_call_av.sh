#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
NF_AB=none timeout 600 python tools/glow_leaf_ab.py 2>&1 | grep -v Warn | tail -1
timeout 900 python bench.py > gpurun_out/r6av_bench.json 2> gpurun_out/r6av_bench.err; echo "bench rc=$?"
