#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "conv or glow or Glow or captures_into_one_graph or made or maf or inv1x1 or Invertible or channel_sum" 2>&1 | tail -3
for i in 1 2; do
echo "c4: $(timeout 300 python tools/config_bench.py 4 2>&1 | tail -1 | cut -c150-330)"
done
echo "c5: $(timeout 300 python tools/config_bench.py 5 2>&1 | tail -1 | cut -c1-330)"
