set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name counters... -- cmd
  name=$1; shift; ctr=$1; shift
  timeout 170 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r2k/$name -- "$@" > $R/gpurun_out/r2k/$name.log 2>&1
  echo "$name rc=$?"
}
mkdir -p $R/gpurun_out/r2k
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --no-graph --no-secondary"
run bench_fetch "FETCH_SIZE" $B
run bench_write "WRITE_SIZE" $B
run bench_sq "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" $B
G="python $R/tools/config_bench.py 4"
run glow_fetch "FETCH_SIZE" $G
run glow_write "WRITE_SIZE" $G
run glow_sq "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" $G
