#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for i in 1 2 3; do
echo "new: $(timeout 300 python tools/maf_inverse_bench.py --reps 20 2>&1 | tail -1 | cut -c1-200)"
echo "old: $(NF_MI355X_LIB=$V/mafold.so timeout 300 python tools/maf_inverse_bench.py --reps 20 2>&1 | tail -1 | cut -c1-200)"
done
