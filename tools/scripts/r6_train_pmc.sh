# round 6: counter passes of the training step (flat parameters, pair path): HBM bytes and MFMA busy per kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/train_$n -- python $R/tools/train_bench.py --steps 3 --flat > $R/gpurun_out/pmc/train_$n.log 2>&1
  echo "$n rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc train "python tools/train_bench.py --steps 3 --flat" > gpurun_out/pmc/r06_train_step_pmc.json 2> gpurun_out/pmc/summ.log
rm -rf gpurun_out/pmc/train_FETCH_SIZE gpurun_out/pmc/train_WRITE_SIZE gpurun_out/pmc/train_SQ_VALU_MFMA_BUSY_CYCLES
head -c 3000 gpurun_out/pmc/r06_train_step_pmc.json
