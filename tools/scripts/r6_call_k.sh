#!/bin/bash
# Round 6 call k: asm LDS-DMA + operand look-ahead in the headline kernel and final_bwd: parity + A/B
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/normalizing-flows_amd/lib/variants
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hygiene.py tests/test_gpu_training.py -x -q -k "fused or chain or counted_waits or final or benchmark_shape or one_call or pair" 2>&1 | tail -3
for i in 1 2; do
for v in "" pf_fence pf_old pf_nopf; do
echo "--- ${v:-new}"
if [ -n "$v" ]; then export NF_MI355X_LIB=$V/$v.so; else unset NF_MI355X_LIB; fi
timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-300
done
done
