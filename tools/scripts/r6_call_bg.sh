#!/bin/bash
cd $GRAFT_REPO_ROOT
NF_MADE_TR128=0 timeout 300 python tools/made_tr128_ab.py 2>&1 | grep "^{"
NF_MADE_TR128=1 timeout 300 python tools/made_tr128_ab.py 2>&1 | grep "^{\|Error\|error" | head -5
NF_MADE_TR128=0 timeout 300 python tools/made_tr128_ab.py 2>&1 | grep "^{"
