#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/glow_copy_audit.py 2>&1 | grep -v Warn | tail -64
