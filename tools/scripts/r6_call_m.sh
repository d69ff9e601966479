#!/bin/bash
# Round 6 call m: kernel profile of the Glow (config 4) training step as it stands
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/glow_train_prof -o glow -- python tools/glow_train_bench.py --steps 5 > gpurun_out/r6m_bench.log 2>&1
tail -3 gpurun_out/r6m_bench.log
f=$(find gpurun_out/glow_train_prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_glow_train_kernel_stats.csv
find gpurun_out/glow_train_prof -type f ! -name "*kernel_stats.csv" -delete
head -45 gpurun_out/r06_glow_train_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
