R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2o
for v in pair nopair; do
  if [ $v = nopair ]; then export NF_MI355X_LIB=$R/normalizing-flows_amd/lib/variants/mafnopair.so; else unset NF_MI355X_LIB; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 170 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r2o/${v}_$c -- python $R/tools/config_bench.py 5 > $R/gpurun_out/r2o/${v}_$c.log 2>&1
    echo "$v $c rc=$?"
  done
done
