#!/bin/bash
# Round 6 call j: Glow inference A/B (asm DMA / look-ahead products / per-block blob pointers) + parity
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hygiene.py -x -q -k "glow or counted_waits" 2>&1 | tail -3
for i in 1 2; do
echo "--- new"; timeout 300 python tools/config_bench.py 4 2>&1 | tail -1 | cut -c1-600
for v in gc_old gc_nopipe gc_builtin; do
echo "--- $v"; NF_MI355X_LIB=$V/$v.so timeout 300 python tools/config_bench.py 4 2>&1 | tail -1 | cut -c1-600
done
done
