#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/glow_train_prof
rm -rf $O
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $GRAFT_REPO_ROOT/tools/glow_train_bench.py > /dev/null 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/glow_train_prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r06_glow_train_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
head -3 $f | cut -c1-100
