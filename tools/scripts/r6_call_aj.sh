#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for i in 1 2; do
for v in "" setprio schedpipe schedpipe3; do
if [ -n "$v" ]; then export NF_MI355X_LIB=$V/$v.so; else unset NF_MI355X_LIB; fi
echo "--- ${v:-base} $(timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))")"
done
done
