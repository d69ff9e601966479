#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in soak_r5.py soak_r6.py; do
  echo "== $t"; (timeout 600 python tools/$t) 2>&1 | grep -v Warn | tail -4 | cut -c1-600
done
