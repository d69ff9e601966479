#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for b in 65600 65537; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6ab_$b -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --flat --batch $b > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r6ab_$b -name "*kernel_stats.csv" | head -1); echo "== B=$b"; head -14 $f | cut -d, -f1-4 | cut -c1-130
find $GRAFT_REPO_ROOT/gpurun_out/r6ab_$b -type f -delete
done
