#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for i in 1 2; do
for v in "" rl8 rl4; do
if [ -n "$v" ]; then export NF_MI355X_LIB=$V/$v.so; else unset NF_MI355X_LIB; fi
echo "${v:-rl16}: $(timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-120)"
done
done
unset NF_MI355X_LIB
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3
