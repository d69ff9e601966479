#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
echo "== nw4: parity"; NF_MI355X_LIB=$V/nw4.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or chain or golden or model_c2" 2>&1 | tail -3
cat > /tmp/sw.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import normflows_amd as nfa
from bench import build_c2_model
m = build_c2_model().to("cuda:0")
with torch.no_grad():
    for B in (1024, 4096, 16384, 32768, 49152, 65536, 131072):
        x = torch.randn(B, 64, device="cuda:0")
        for _ in range(3): m.log_prob(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.log_prob(x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
        xs = m.sample(B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.sample(B)
        torch.cuda.synchronize(); ms2 = (time.perf_counter() - t0) * 100
        print("B=%6d log_prob %.3f ms (%.2f M rows/s)  sample %.3f ms" % (B, ms, B / ms / 1e3, ms2))
PY
echo "== nw8"; timeout 300 python /tmp/sw.py 2>&1 | grep "B="
echo "== nw4"; NF_MI355X_LIB=$V/nw4.so timeout 300 python /tmp/sw.py 2>&1 | grep "B="
