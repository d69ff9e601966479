#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
echo "async: $(timeout 300 python tools/train_bench.py --steps 10 --flat 2>&1 | tail -1 | cut -c1-60)"
echo "sync : $(timeout 300 python tools/train_bench.py --steps 10 --flat --no-async 2>&1 | tail -1 | cut -c1-60)"
done
