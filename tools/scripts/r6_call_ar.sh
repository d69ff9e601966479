#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/maf_pos_prof
rm -rf $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $GRAFT_REPO_ROOT/tools/maf_wgrad_pos_ab.py > /dev/null 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/maf_pos_prof -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-160
