#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/soak_r6.py --json gpurun_out/r06_soak.json 2>&1 | grep -v Warn | tail -80
