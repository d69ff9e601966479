cd /tmp; export TMPDIR=/tmp
for cfg in ${MW_CFGS:-0:1536 1:1536 0:2048 1:2048 0:3072 1:3072 1:4096 1:6144}; do
  set -- $(echo $cfg | tr : " ")
  export NF_MW_TILE_MAJOR=$1 NF_MW_SLOTS=$2
  rm -rf /tmp/mwp; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mwp -- python $GRAFT_REPO_ROOT/tools/made_train_bench.py --only > /tmp/mw.log 2>&1
  f=$(find /tmp/mwp -name "*kernel_stats.csv" | head -1)
  echo "tile_major=$1 slots=$2: $(grep -h hand_written /tmp/mw.log | cut -c1-40) $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'made_' in r['Name']: print(r['Name'].split('(')[0].split('::')[-1][:24], r['AverageNs'][:8], end=' | ')
")"
done
