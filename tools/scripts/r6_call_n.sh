#!/bin/bash
# Round 6 call n: zero-scratch tile-engine / MAF kernels: parity + numbers
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -x -q -m gpu -k "wide or made or maf or arnsf or autoregressive or bins or glow or convnet" 2>&1 | tail -3
timeout 600 python tools/wide_bench.py --json gpurun_out/r6n_wide.json 2>&1 | tail -12; timeout 300 python tools/wide_bench.py --bins 16 2>&1 | tail -8
for i in 1 2; do timeout 600 python tools/config_bench.py 5 2>&1 | tail -1 | cut -c1-400; done
