#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6y_stats -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --flat > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6y_stats -name "*kernel_stats.csv" | head -1); head -10 $f | cut -d, -f1-4 | cut -c1-120
find gpurun_out/r6y_stats -type f ! -name "*kernel_stats.csv" -delete
