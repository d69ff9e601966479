#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/batch_sweep.py --json gpurun_out/r06_batch_sweep.json 2>&1 | grep rows
