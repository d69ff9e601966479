#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 65536 65537 65600 1000; do
echo "B=$b: $(timeout 300 python tools/train_bench.py --steps 5 --flat --batch $b 2>&1 | tail -1 | cut -c1-90)"
done
