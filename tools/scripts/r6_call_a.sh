# round 6, call A: baseline of the training step on this round's box + phase traces of the two backward kernels
mkdir -p gpurun_out/a
timeout 300 python tools/train_bench.py --steps 5 --fused-adam > gpurun_out/a/train.log 2>&1; tail -3 gpurun_out/a/train.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/a/train_stats -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --fused-adam > $GRAFT_REPO_ROOT/gpurun_out/a/train_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/a/train_stats -name "*kernel_stats.csv" | head -1) gpurun_out/a/train_kernel_stats.csv; rm -rf gpurun_out/a/train_stats
head -14 gpurun_out/a/train_kernel_stats.csv | cut -c1-160
NF_MI355X_LIB=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants/bbtrace.so timeout 200 python tools/resblock_trace.py > gpurun_out/a/bbtrace.log 2>&1; cat gpurun_out/a/bbtrace.log | cut -c1-400
NF_MI355X_LIB=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants/fbtrace.so timeout 200 python tools/final_bwd_probe.py --trace > gpurun_out/a/fbtrace.log 2>&1; cat gpurun_out/a/fbtrace.log | cut -c1-300
