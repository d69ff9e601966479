# rocprofv3 counter passes of one training run (tools/train_bench.py): FETCH_SIZE, WRITE_SIZE and the MFMA-busy set, each in its
# own pass with --kernel-trace only (gpurun refuses counter runs combined with the sys / hip trace domains).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2x
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r2x/train_$n -- python $R/tools/train_bench.py --steps 3 --fused-adam > $R/gpurun_out/r2x/train_$n.log 2>&1
  echo "$n rc=$?"
done
# ... and the per-kernel time table of the same run (no counters)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2x/train_stats -- python $R/tools/train_bench.py --steps 5 --fused-adam > $R/gpurun_out/r2x/train_stats.log 2>&1
echo "stats rc=$?"; tail -1 $R/gpurun_out/r2x/train_stats.log
