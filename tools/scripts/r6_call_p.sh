#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
echo "== new (rqs_regs_h)"; timeout 300 python tools/wide_row_diag.py 2>&1 | grep -v Warn | tail -12
echo "== full knots (rqs_regs_t)"; NF_MI355X_LIB=$V/fullknots.so timeout 300 python tools/wide_row_diag.py 2>&1 | grep -v Warn | tail -12
