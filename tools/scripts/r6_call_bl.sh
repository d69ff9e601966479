#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/soak_made_r6.py 2>&1 | grep "^{\|Error\|error" | tail -3
