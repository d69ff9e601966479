#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "side_stream or flat_parameters or pair or one_call or benchmark_shape" 2>&1 | tail -2
for i in 1 2 3; do
echo "async: $(timeout 300 python tools/train_bench.py --steps 10 --flat 2>&1 | tail -1 | cut -c1-60)"
echo "sync : $(timeout 300 python tools/train_bench.py --steps 10 --flat --no-async 2>&1 | tail -1 | cut -c1-60)"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6x_stats -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --flat > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6x_stats -name "*kernel_stats.csv" | head -1); head -10 $f | cut -d, -f1-4 | cut -c1-120
find gpurun_out/r6x_stats -type f ! -name "*kernel_stats.csv" -delete
