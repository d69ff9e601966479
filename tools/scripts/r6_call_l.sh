#!/bin/bash
# Round 6 call l: asm LDS-DMA + look-ahead as built: parity of every fused path + the headline / sample / training numbers
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hygiene.py tests/test_gpu_training.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
timeout 300 python tools/train_bench.py --steps 8 --flat 2>&1 | tail -1 | cut -c1-300
done
