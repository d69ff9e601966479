V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
for v in "" fbnorows fbnoident; do
  if [ -z "$v" ]; then timeout 200 python tools/final_bwd_probe.py 2>&1 | tail -1; else NF_MI355X_LIB=$V/$v.so timeout 200 python tools/final_bwd_probe.py 2>&1 | tail -1; fi
done
