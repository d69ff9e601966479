# round 6, call D: the pair training path (LU fused into the training forward, composed LU backward): tests, A/B, kernel stats
mkdir -p gpurun_out/d
timeout 1200 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/d/pytest_train.log 2>&1; tail -5 gpurun_out/d/pytest_train.log | cut -c1-400
for i in 1 2; do
timeout 300 python tools/train_bench.py --steps 8 --flat > gpurun_out/d/train.log 2>&1; echo "pair: $(tail -1 gpurun_out/d/train.log | cut -c1-200)"
timeout 300 python tools/train_bench.py --steps 8 --flat --no-pair > gpurun_out/d/train_nopair.log 2>&1; echo "no pair: $(tail -1 gpurun_out/d/train_nopair.log | cut -c1-200)"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/d/train_stats -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --flat > $GRAFT_REPO_ROOT/gpurun_out/d/train_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/d/train_stats -name "*kernel_stats.csv" | head -1) gpurun_out/d/train_kernel_stats.csv; rm -rf gpurun_out/d/train_stats
head -14 gpurun_out/d/train_kernel_stats.csv | cut -c1-150
