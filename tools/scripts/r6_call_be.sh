#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
