#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/wide_train_bench.py 2>&1 | grep -v Warn | tail -6 | cut -c1-400
