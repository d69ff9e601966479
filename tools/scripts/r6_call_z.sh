#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
