#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/maf_solve_fast_ab.py 2>&1 | grep -v Warn | tail -3
