#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
timeout 900 python -m pytest tests -m gpu -x -q -k "conv or glow or Glow or captures_into_one_graph" 2>&1 | tail -3
for i in 1 2; do
echo "new: $(timeout 300 python tools/config_bench.py 4 2>&1 | tail -1 | cut -c150-330)"
echo "old: $(NF_MI355X_LIB=$V/convold.so timeout 300 python tools/config_bench.py 4 2>&1 | tail -1 | cut -c150-330)"
done
