# round 6, call B: the one-call layer backward + flat parameters: tests, A/B timings, kernel stats
mkdir -p gpurun_out/b
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "benchmark_shape or one_call or flat_parameters or fused_final_backward or prepack or captures_into_one_graph or follows_fused" > gpurun_out/b/pytest_train.log 2>&1; tail -5 gpurun_out/b/pytest_train.log | cut -c1-300
for args in "--fused-adam --no-onecall" "--fused-adam" "--flat"; do
  timeout 300 python tools/train_bench.py --steps 8 $args > gpurun_out/b/train.log 2>&1; echo "$args: $(tail -1 gpurun_out/b/train.log | cut -c1-200)"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/b/train_stats -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 --flat > $GRAFT_REPO_ROOT/gpurun_out/b/train_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/b/train_stats -name "*kernel_stats.csv" | head -1) gpurun_out/b/train_kernel_stats.csv; rm -rf gpurun_out/b/train_stats
head -16 gpurun_out/b/train_kernel_stats.csv | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -k "train_mode" > gpurun_out/b/pytest_bench.log 2>&1; tail -3 gpurun_out/b/pytest_bench.log | cut -c1-300
