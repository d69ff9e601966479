#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for i in 1 2; do
timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-200
NF_MI355X_LIB=$V/noreduce.so timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-200
done
